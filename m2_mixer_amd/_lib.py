"""ctypes binding of libm2mixer.so (include/m2mixer.h).

This is the reference-side binding a maintainer would add: plain pointers and
sizes across the boundary, torch only supplies device memory (`data_ptr()`)
and the stream (`torch.cuda.current_stream().cuda_stream`).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets a RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M2M_LIB_PATH", os.path.join(_HERE, "libm2mixer.so"))   # override: diagnostic builds
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 17
MAX_BLOCKS = 8
ROWS_PER_WG = 16
HCHN_PAD = 4096          # >= the pad between operand streams the library uses (csrc/tile.h M2M_HCHN_PAD)
SPLIT_ROWS = 128         # token rows of a column-split channel workgroup (csrc/split.h SP_ROWS)
SPLIT_MAX = 8            # column splits (slabs)
SPLIT_GPART = 1472       # floats per workgroup and launch of the small-gradient partial buffer (M2M_SPLIT_GPART)
PREC_BF16, PREC_F32 = 0, 1
PREC_BY_NAME = {"bf16": PREC_BF16, "fp32": PREC_F32, "f32": PREC_F32}

_fp = C.c_void_p  # every device pointer crosses the boundary as void*


class Block(C.Structure):
    """m2m_block"""
    PARAMS = ["ln1_w", "ln1_b", "tok_w1", "tok_b1", "tok_w2", "tok_b2", "ln2_w", "ln2_b",
              "ch_w1", "ch_b1", "ch_w2", "ch_b2"]
    PACKED = ["w1n", "w2c", "w2tn", "w1tc", "ch_b1p"]
    GRADS = ["g_" + p for p in PARAMS]
    SAVED = ["x_in", "x_mid", "at_chn", "dyt_chn", "h_chn", "dh_chn"]
    _fields_ = [(n, _fp) for n in PARAMS + PACKED + GRADS + SAVED]


class Tower(C.Structure):
    """m2m_tower"""
    _fields_ = [("prec", C.c_int32), ("D", C.c_int32), ("N", C.c_int32), ("T", C.c_int32), ("C", C.c_int32),
                ("Cp", C.c_int32), ("nblocks", C.c_int32), ("has_final_ln", C.c_int32),
                ("p_drop", C.c_float), ("site_base", C.c_uint32),
                ("lnf_w", _fp), ("lnf_b", _fp), ("g_lnf_w", _fp), ("g_lnf_b", _fp), ("x_final", _fp),
                ("ws_a", _fp), ("ws_b", _fp), ("blk", Block * MAX_BLOCKS),
                # split path (csrc/split.h): slab buffer of the column-split launches, carry stream, per-block operand images
                ("slabs", _fp), ("nsplit", C.c_int32), ("wgrad_flags", C.c_int32), ("xres", _fp), ("gpart", _fp),
                ("a_nat", _fp * MAX_BLOCKS), ("dy_nat", _fp * MAX_BLOCKS),
                # weight-gradient slot: second row group of a long tower, folded in by the optimizer
                ("wslot", _fp * MAX_BLOCKS),
                # packed image of d_x0^T for the patch-embedding weight gradient
                ("dx0_chn", _fp)]


WGRAD_OVERWRITE = 1      # Tower.wgrad_flags (M2M_WGRAD_OVERWRITE)
WGRAD_REDUCES_SMALL = 2  # Tower.wgrad_flags (M2M_WGRAD_REDUCES_SMALL)
WGRAD_GROUP_SLOTS = 4    # Tower.wgrad_flags (M2M_WGRAD_GROUP_SLOTS)
MAX_GRAD_RANGES = 16


class GradRange(C.Structure):
    """m2m_grad_range"""
    _fields_ = [("lo", C.c_int64), ("n", C.c_int64), ("add", _fp), ("keep", C.c_int32), ("reserved", C.c_int32)]


class TowerIO(C.Structure):
    """m2m_tower_io"""
    _fields_ = [("x0", _fp), ("x0_ss", C.c_int64), ("out", _fp), ("out_ss", C.c_int64), ("pooled", _fp),
                ("x0_parts", C.c_int32), ("x0_part_stride", C.c_int64)]


class StepHead(C.Structure):
    """m2m_step_head"""
    _fields_ = [("adam_state", _fp), ("drop_counter", _fp), ("losses", _fp), ("nlosses", C.c_int32)]


class TowerGIO(C.Structure):
    """m2m_tower_gio"""
    _fields_ = [("d_out", _fp), ("d_out_ss", C.c_int64), ("d_pooled", _fp), ("d_x0", _fp), ("d_x0_ss", C.c_int64)]


class Embed(C.Structure):
    """m2m_embed"""
    _fields_ = [("prec", C.c_int32), ("Cin", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("ph", C.c_int32),
                ("pw", C.c_int32), ("D", C.c_int32), ("K", C.c_int32), ("Kp", C.c_int32),
                ("w", _fp), ("b", _fp), ("wn", _fp), ("g_w", _fp), ("g_b", _fp), ("wgrad_flags", C.c_int32), ("reserved", C.c_int32)]


class Head(C.Structure):
    """m2m_head"""
    _fields_ = [("pooled", _fp), ("w", _fp), ("b", _fp), ("g_w", _fp), ("g_b", _fp), ("d_pooled", _fp),
                ("weight", C.c_float), ("g_part", _fp), ("tokens", _fp), ("tok_sample_stride", C.c_int64), ("ntok", C.c_int32)]


MLP_MAX_LAYERS = 4


class Mlp(C.Structure):
    """m2m_mlp"""
    _fields_ = [("nlayers", C.c_int32), ("has_out", C.c_int32), ("dims", C.c_int32 * (MLP_MAX_LAYERS + 1)),
                ("p_drop", C.c_float), ("site_base", C.c_uint32),
                ("w", _fp * MLP_MAX_LAYERS), ("b", _fp * MLP_MAX_LAYERS), ("g_w", _fp * MLP_MAX_LAYERS),
                ("g_b", _fp * MLP_MAX_LAYERS), ("act", _fp * MLP_MAX_LAYERS)]


# name -> (restype, argtypes); every symbol include/m2mixer.h declares
SIGNATURES = {
    "m2m_abi_version": (C.c_int, []),
    "m2m_last_error": (C.c_char_p, []),
    "m2m_packed_bytes": (C.c_int64, [C.c_int, C.c_int64, C.c_int64]),
    "m2m_pack": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _fp, _fp]),
    "m2m_pack_tower": (C.c_int, [C.POINTER(Tower), _fp]),
    "m2m_pack_embed": (C.c_int, [C.POINTER(Embed), _fp]),
    "m2m_pack_skips_w1tc": (C.c_int, [C.POINTER(Tower)]),
    "m2m_pack_all": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.c_int, C.POINTER(C.POINTER(Embed)), C.c_int, _fp]),
    "m2m_embed_forward": (C.c_int, [C.POINTER(Embed), _fp, C.c_int, _fp, _fp]),
    "m2m_embed_forward_head": (C.c_int, [C.POINTER(Embed), _fp, C.c_int, _fp, C.POINTER(StepHead), _fp]),
    "m2m_tower_forward": (C.c_int, [C.POINTER(Tower), _fp, C.c_int64, C.c_int, _fp, C.c_int64, _fp, C.c_int,
                                    C.c_uint32, C.c_uint32, _fp, _fp]),
    "m2m_tower_backward": (C.c_int, [C.POINTER(Tower), C.c_int, _fp, C.c_int64, _fp, _fp, C.c_int64,
                                     C.c_uint32, C.c_uint32, _fp, _fp]),
    "m2m_tower_backward_heads_ok": (C.c_int, [C.POINTER(Tower), C.c_int, C.c_int, C.c_int]),
    "m2m_tower_backward_heads": (C.c_int, [C.POINTER(Tower), C.c_int, C.POINTER(Head), C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp,
                                           C.c_int64, C.c_uint32, C.c_uint32, _fp, _fp]),
    "m2m_tower_wgrad": (C.c_int, [C.POINTER(Tower), C.c_int, C.c_uint32, C.c_uint32, _fp, _fp]),
    "m2m_towers_can_group": (C.c_int, [C.POINTER(Tower), C.POINTER(Tower), C.c_int]),
    "m2m_towers_forward": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.POINTER(TowerIO), C.c_int, C.c_int, C.c_int, C.c_uint32,
                                     C.c_uint32, _fp, _fp]),
    "m2m_towers_forward_embeds_ok": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.c_int, C.POINTER(C.POINTER(Embed)), C.c_int]),
    "m2m_towers_forward_embeds": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.POINTER(TowerIO), C.c_int, C.POINTER(C.POINTER(Embed)),
                                            C.POINTER(_fp), C.POINTER(StepHead), C.c_int, C.c_int, C.c_uint32, C.c_uint32, _fp, _fp]),
    "m2m_towers_backward": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.POINTER(TowerGIO), C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                      _fp, _fp]),
    "m2m_towers_wgrad": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.POINTER(_fp), C.c_int, C.POINTER(C.POINTER(Embed)), C.POINTER(_fp),
                                   C.POINTER(_fp), C.POINTER(C.POINTER(Tower)), C.c_int, C.c_int, C.c_uint32, C.c_uint32, _fp, _fp]),
    "m2m_towers_wgrad_heads": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.POINTER(_fp), C.c_int, C.POINTER(C.POINTER(Embed)), C.POINTER(_fp),
                                         C.POINTER(_fp), C.POINTER(C.POINTER(Tower)), C.c_int, C.c_int, C.c_uint32, C.c_uint32, _fp,
                                         C.POINTER(Head), C.c_int, C.c_int, _fp]),
    "m2m_towers_wgrad_tail": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.POINTER(_fp), C.c_int, C.POINTER(C.POINTER(Embed)), C.POINTER(_fp),
                                        C.POINTER(_fp), C.POINTER(C.POINTER(Tower)), C.c_int, C.c_int, C.c_uint32, C.c_uint32, _fp,
                                        C.POINTER(Head), C.c_int, C.c_int, _fp, _fp]),
    "m2m_heads_part_tiles": (C.c_int, [C.c_int]),
    "m2m_wgrad_form": (C.c_int, [C.POINTER(Tower), C.c_int]),
    "m2m_embeds_wgrad_form": (C.c_int, [C.POINTER(C.POINTER(Embed)), C.POINTER(C.POINTER(Tower)), C.c_int, C.c_int]),
    "m2m_wgrad_groups": (C.c_int, [C.POINTER(Tower), C.c_int]),
    "m2m_wgrad_slot_groups": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.c_int, C.c_int]),
    "m2m_wgrad_fold": (C.c_int, [C.POINTER(Tower), _fp]),
    "m2m_adam_step_ranges": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int64, _fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_int, C.POINTER(GradRange), C.c_int, _fp]),
    "m2m_counter_add": (C.c_int, [_fp, C.c_uint32, _fp]),
    "m2m_embed_wgrad": (C.c_int, [C.POINTER(Embed), _fp, _fp, C.c_int, _fp]),
    "m2m_embeds_wgrad": (C.c_int, [C.POINTER(C.POINTER(Embed)), C.POINTER(_fp), C.POINTER(_fp), C.c_int, C.c_int, _fp]),
    "m2m_embeds_forward": (C.c_int, [C.POINTER(C.POINTER(Embed)), C.POINTER(_fp), C.POINTER(_fp), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                     C.c_int, C.c_int, C.POINTER(StepHead), _fp]),
    "m2m_embed_fwd_splits": (C.c_int, [C.POINTER(Embed)]),
    "m2m_heads_ce": (C.c_int, [C.POINTER(Head), C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int, _fp]),
    "m2m_step_prologue": (C.c_int, [_fp, _fp, _fp, C.c_int, _fp]),
    "m2m_heads_bce": (C.c_int, [C.POINTER(Head), C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int, _fp]),
    "m2m_mlp_forward": (C.c_int, [C.POINTER(Mlp), _fp, C.c_int, _fp, C.c_int64, _fp, C.c_int, C.c_uint32, C.c_uint32,
                                  _fp, _fp]),
    "m2m_mlp_backward": (C.c_int, [C.POINTER(Mlp), _fp, C.c_int, _fp, C.c_int64, _fp, _fp]),
    "m2m_mlp_forward_ride": (C.c_int, [C.POINTER(Mlp), _fp, C.c_int, _fp, C.c_int64, _fp, C.c_int, C.c_uint32, C.c_uint32, _fp]),
    "m2m_mlp_backward_ride": (C.c_int, [C.POINTER(Mlp), _fp, C.c_int, _fp, C.c_int64, _fp]),
    "m2m_mlp_ride_flush": (C.c_int, [_fp]),
    "m2m_adam_step": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, _fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_int, _fp]),
    "m2m_adam_step_bf16": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int64, _fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_int, _fp]),
    "m2m_adam_pack_plan_bytes": (C.c_int64, []),
    "m2m_adam_pack_plan": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.c_int, C.POINTER(C.POINTER(Embed)), C.c_int, _fp, _fp, _fp, _fp, _fp,
                                    C.c_int64, _fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _fp]),
    "m2m_adam_pack_plan_ranges": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.c_int, C.POINTER(C.POINTER(Embed)), C.c_int, _fp, _fp, _fp, _fp, _fp,
                                           C.c_int64, _fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                           C.POINTER(GradRange), C.c_int, _fp]),
    "m2m_adam_pack_all": (C.c_int, [C.POINTER(C.POINTER(Tower)), C.c_int, C.POINTER(C.POINTER(Embed)), C.c_int, _fp, _fp, _fp]),
    "m2m_dropout_mask": (C.c_int, [C.POINTER(Tower), C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _fp, _fp]),
    "m2m_gelu_probe": (C.c_int, [_fp, _fp, _fp, C.c_int64, _fp]),
    "m2m_gemm_probe": (C.c_int, [C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp]),
    "m2m_clock_probe": (C.c_int, [_fp, C.c_int, C.c_int, _fp]),
}

_lib = None


def csrc_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, the C header): identifies the code a profile was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                    [os.path.join(os.path.dirname(_HERE), "include", "m2mixer.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libm2mixer.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(["make", "-C", CSRC, "-j", str(min(6, os.cpu_count() or 2))], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libm2mixer.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the M2-Mixer hot path exists only as the HIP library "
                "(python -c 'import __graft_entry__ as g; g.build()' builds it); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = L.m2m_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libm2mixer ABI {got} != binding ABI {ABI_VERSION}; rebuild")
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().m2m_last_error().decode("utf-8", "replace")
        kind = "unsupported shape/argument" if rc == -1 else "HIP runtime error"
        raise RuntimeError(f"libm2mixer {what}: {kind}: {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def packed_bytes(prec: int, I: int, K: int) -> int:
    kb = 32 if prec == PREC_BF16 else 16
    return ((I + 15) // 16) * ((K + kb - 1) // kb) * 1024
