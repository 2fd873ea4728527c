"""Host-side state of the HIP towers: descriptor structs, packed weight copies, saved-activation
buffers and the launches.  Pure plumbing -- every FLOP of the hot path is in libm2mixer.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L

# block parameter order used everywhere on the host side; values are the reference's state-dict keys
# relative to a MixerBlock
BLOCK_KEYS = {
    "ln1_w": "token_mix.0.weight", "ln1_b": "token_mix.0.bias",
    "tok_w1": "token_mix.2.net.0.weight", "tok_b1": "token_mix.2.net.0.bias",
    "tok_w2": "token_mix.2.net.3.weight", "tok_b2": "token_mix.2.net.3.bias",
    "ln2_w": "channel_mix.0.weight", "ln2_b": "channel_mix.0.bias",
    "ch_w1": "channel_mix.1.net.0.weight", "ch_b1": "channel_mix.1.net.0.bias",
    "ch_w2": "channel_mix.1.net.3.weight", "ch_b2": "channel_mix.1.net.3.bias",
}
BLOCK_FIELDS = list(BLOCK_KEYS.keys())


def block_param_shapes(D: int, N: int, T: int, Cc: int) -> Dict[str, tuple]:
    return {"ln1_w": (D,), "ln1_b": (D,), "tok_w1": (T, N), "tok_b1": (T,), "tok_w2": (N, T), "tok_b2": (N,),
            "ln2_w": (D,), "ln2_b": (D,), "ch_w1": (Cc, D), "ch_b1": (Cc,), "ch_w2": (D, Cc), "ch_b2": (D,)}


def _check_tensor(t: torch.Tensor, shape, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: m2_mixer_amd runs on the GPU only (got a {t.device} tensor); "
                           "there is no CPU path -- the CPU restatement under oracle/ is test infrastructure")
    if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected contiguous float32 {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")


class TowerRuntime:
    """One m2m_tower: `nblocks` MixerBlocks (+ final LayerNorm) over (B, N, D) tokens."""

    def __init__(self, D: int, N: int, T: int, Cc: int, nblocks: int, has_final_ln: bool, p_drop: float,
                 prec: int, site_base: int = 0):
        if nblocks > L.MAX_BLOCKS:
            raise RuntimeError(f"one m2m_tower holds at most {L.MAX_BLOCKS} blocks; deeper towers are a chain of them "
                               "(modules/mixer.py does that; the fused engines take towers of up to 8 blocks)")
        self.D, self.N, self.T, self.C, self.nblocks = D, N, T, Cc, nblocks
        self.Cp = (Cc + 31) // 32 * 32
        self.prec = prec
        self.has_final_ln = bool(has_final_ln)
        self.desc = L.Tower()
        d = self.desc
        d.prec, d.D, d.N, d.T, d.C, d.Cp = prec, D, N, T, Cc, self.Cp
        d.nblocks, d.has_final_ln = nblocks, int(self.has_final_ln)
        d.p_drop = float(p_drop)
        d.site_base = site_base
        self._keep: Dict[str, object] = {}      # keeps every tensor whose pointer sits in the descriptor alive
        self._param_ptrs: List[int] = []
        self._param_versions: List[int] = []
        self._packed_for: Optional[List[int]] = None
        self._bufB = 0
        self._wsB = 0
        self._splitB = 0
        self.device = None
        # wide path (include/m2mixer.h): token mixing and channel mixing are separate launches
        self.wide = N > 8 or D > 128

    # ---- parameters --------------------------------------------------------------------------------
    def bind_params(self, blocks: Sequence[Dict[str, torch.Tensor]], lnf: Optional[Sequence[torch.Tensor]]):
        """blocks[i][field] for field in BLOCK_FIELDS; lnf = (weight, bias) of the final LayerNorm."""
        shapes = block_param_shapes(self.D, self.N, self.T, self.C)
        ptrs, vers = [], []
        for i, bp in enumerate(blocks):
            for f in BLOCK_FIELDS:
                t = bp[f]
                _check_tensor(t, shapes[f], f"block {i} {BLOCK_KEYS[f]}")
                setattr(self.desc.blk[i], f, t.data_ptr())
                ptrs.append(t.data_ptr())
                vers.append(t._version)
            self._keep[f"params{i}"] = dict(bp)
        if self.has_final_ln:
            for t, f in zip(lnf, ("lnf_w", "lnf_b")):
                _check_tensor(t, (self.D,), f"layer_norm {f}")
                setattr(self.desc, f, t.data_ptr())
                ptrs.append(t.data_ptr())
                vers.append(t._version)
            self._keep["lnf"] = tuple(lnf)
        self.device = blocks[0]["ln1_w"].device if blocks else lnf[0].device
        self._param_ptrs, self._param_versions = ptrs, vers
        self._packed_for = None            # new storage: whatever the packed copies hold belongs to the old one
        self._ensure_packed_buffers()

    def params_changed(self, blocks, lnf) -> bool:
        """True when any parameter storage moved (rebind needed)."""
        ptrs = [bp[f].data_ptr() for bp in blocks for f in BLOCK_FIELDS]
        if self.has_final_ln:
            ptrs += [lnf[0].data_ptr(), lnf[1].data_ptr()]
        return ptrs != self._param_ptrs

    def _ensure_packed_buffers(self):
        if "packed0" in self._keep or self.nblocks == 0:
            return
        nb = L.packed_bytes(self.prec, self.Cp, self.D)
        for i in range(self.nblocks):
            bufs = {k: torch.zeros(nb, dtype=torch.uint8, device=self.device) for k in ("w1n", "w2c", "w2tn", "w1tc")}
            bufs["ch_b1p"] = torch.zeros(self.Cp, dtype=torch.float32, device=self.device)
            for k, v in bufs.items():
                setattr(self.desc.blk[i], k, v.data_ptr())
            self._keep[f"packed{i}"] = bufs

    def _pack_versions(self):
        cur = []
        for i in range(self.nblocks):
            bp = self._keep[f"params{i}"]
            cur += [bp["ch_w1"]._version, bp["ch_w2"]._version, bp["ch_b1"]._version]
        return cur

    def pack_all_skips_w1tc(self) -> bool:
        """pack_all leaves this tower's w1tc copies unwritten (nothing reads them: m2m_pack_skips_w1tc)."""
        return bool(L.lib().m2m_pack_skips_w1tc(C.byref(self.desc)))

    def mark_packed(self):
        self._packed_for = self._pack_versions()

    def invalidate(self):
        """Call after editing a weight in a way autograd's version counter does not see (`p.data.copy_()`, `p.data = ...`):
        the next pack() then rebuilds the packed copies."""
        self._packed_for = None

    def pack(self, force: bool = False):
        """Refresh the packed MFMA-operand copies when the master weights changed (detected through the tensors' version
        counters and, in bind_params, their storage; `.data` edits need invalidate())."""
        if force or self._pack_versions() != self._packed_for:
            L.check(L.lib().m2m_pack_tower(C.byref(self.desc), L.stream_ptr()), "pack_tower")
            self.mark_packed()

    # ---- activations saved for backward ---------------------------------------------------------------
    def ensure_buffers(self, B: int):
        """(Re)allocate the saved-activation buffers for batch B.  The packed operand images are laid out per
        32-row pair of tiles and zero-filled: a ragged last pair must read as zeros in the weight-gradient pass,
        so they are reallocated whenever B changes."""
        if B == self._bufB:
            return
        if self.wide:
            ntiles = (B * self.N + L.ROWS_PER_WG - 1) // L.ROWS_PER_WG
        else:
            spw = L.ROWS_PER_WG // self.N
            ntiles = (B + spw - 1) // spw
        esz = 2 if self.prec == L.PREC_BF16 else 4
        npairs = (ntiles * L.ROWS_PER_WG + 31) // 32
        img = npairs * 32 * self.D * esz
        M = B * self.N
        for i in range(self.nblocks):
            bufs = {"x_in": torch.empty(M, self.D, device=self.device), "x_mid": torch.empty(M, self.D, device=self.device)}
            for k in ("at_chn", "dyt_chn"):
                bufs[k] = torch.zeros(img, dtype=torch.uint8, device=self.device)
            for k in ("h_chn", "dh_chn"):           # hidden activation / its gradient, transposed: rows x Cp elements
                bufs[k] = torch.zeros((self.Cp // 16) * (npairs * 2048 + L.HCHN_PAD), dtype=torch.uint8, device=self.device)
            for k, v in bufs.items():
                setattr(self.desc.blk[i], k, v.data_ptr())
            self._keep[f"saved{i}"] = bufs
        xf = torch.empty(M, self.D, device=self.device)
        self.desc.x_final = xf.data_ptr()
        self._keep["x_final"] = xf
        if self._keep.get("want_dx0_image"):
            img0 = torch.zeros(img, dtype=torch.uint8, device=self.device)
            self.desc.dx0_chn = img0.data_ptr()
            self._keep["dx0_chn"] = img0
        self._bufB = B

    # ---- re-entrancy of the module path (torch autograd): one set of forward-saved activations per forward --------------
    def fresh_saved(self, B: int) -> dict:
        """A NEW set of the buffers a training forward writes and the matching backward reads (block inputs, post-token-mix
        streams, the final LayerNorm's input; on the split path also the LN2 operand images), and the descriptor pointed at
        it.  The autograd Function keeps the set in its ctx, so a validation forward between a training forward and its
        backward, or gradient accumulation over several micro-batches, no longer overwrites what a pending backward needs.
        Buffers only the backward writes (dYd images, hidden-activation operand streams) stay shared: a backward call is
        atomic."""
        self.ensure_buffers(B)
        M = B * self.N
        f = lambda: torch.empty(M, self.D, device=self.device)
        saved = {"B": B, "x_in": [f() for _ in range(self.nblocks)], "x_mid": [f() for _ in range(self.nblocks)], "x_final": f()}
        if "split" in self._keep and self._splitB == B:
            sp = self._keep["split"]
            saved["a_nat"] = [torch.zeros_like(t) for t in sp["a_nat"]]
            saved["at_chn"] = [torch.zeros_like(self._keep[f"saved{i}"]["at_chn"]) for i in range(self.nblocks)]
        self.use_saved(saved)
        return saved

    def use_saved(self, saved: dict):
        for i in range(self.nblocks):
            self.desc.blk[i].x_in, self.desc.blk[i].x_mid = saved["x_in"][i].data_ptr(), saved["x_mid"][i].data_ptr()
            if "a_nat" in saved:
                self.desc.a_nat[i], self.desc.blk[i].at_chn = saved["a_nat"][i].data_ptr(), saved["at_chn"][i].data_ptr()
        self.desc.x_final = saved["x_final"].data_ptr()
        self._keep["saved_set"] = saved

    def ensure_split(self, B: int):
        """Buffers of the split path (include/m2mixer.h, csrc/split.h): slabs of the column-split channel launches, the carry
        stream, per-block bf16 operand images.  Allocated for every tower the split path can take (fused class, bf16,
        hidden_dim 128); whether a call uses it is the library's decision (batch size; M2M_SPLIT=0/1 overrides)."""
        if self.wide or self.prec != L.PREC_BF16 or self.D != 128 or self.nblocks == 0 or B == self._splitB:
            return
        M = B * self.N
        ntile16 = (M + 15) // 16
        img = ntile16 * (self.D // 32) * 1024
        ntiles = (B + (16 // self.N) - 1) // (16 // self.N)
        bufs = {"slabs": torch.empty(L.SPLIT_MAX, M, self.D, device=self.device), "xres": torch.empty(M, self.D, device=self.device),
                # (+ 3 slot sets: the classification heads of m2m_tower_backward_heads)
                "gpart": torch.zeros((self.nblocks + 4) * ntiles * L.SPLIT_GPART, device=self.device),
                "a_nat": [torch.zeros(img, dtype=torch.uint8, device=self.device) for _ in range(self.nblocks)],
                "dy_nat": [torch.zeros(img, dtype=torch.uint8, device=self.device) for _ in range(self.nblocks)]}
        self.desc.slabs, self.desc.nsplit, self.desc.xres = bufs["slabs"].data_ptr(), L.SPLIT_MAX, bufs["xres"].data_ptr()
        self.desc.gpart = bufs["gpart"].data_ptr()
        for i in range(self.nblocks):
            self.desc.a_nat[i], self.desc.dy_nat[i] = bufs["a_nat"][i].data_ptr(), bufs["dy_nat"][i].data_ptr()
        self._keep["split"] = bufs
        self._splitB = B

    def ensure_workspace(self, B: int):
        """The wide path's two (B*N, D) stream buffers (needed in eval too)."""
        self.ensure_split(B)
        if not self.wide or B == self._wsB:
            return
        ws = torch.empty(2, B * self.N, self.D, device=self.device)
        self.desc.ws_a, self.desc.ws_b = ws[0].data_ptr(), ws[1].data_ptr()
        self._keep["ws"] = ws
        self._wsB = B

    # ---- gradients -----------------------------------------------------------------------------------
    def grad_numel(self) -> int:
        shapes = block_param_shapes(self.D, self.N, self.T, self.C)
        n = sum(int(torch.Size(s).numel()) for s in shapes.values()) * self.nblocks
        return n + (2 * self.D if self.has_final_ln else 0)

    def bind_grads(self, flat: torch.Tensor) -> List[torch.Tensor]:
        """Point every g_* at a slice of `flat` (fp32, >= grad_numel()); returns the views in
        bind_params order (blocks' BLOCK_FIELDS, then lnf weight, bias)."""
        shapes = block_param_shapes(self.D, self.N, self.T, self.C)
        views, off = [], 0
        for i in range(self.nblocks):
            for f in BLOCK_FIELDS:
                n = int(torch.Size(shapes[f]).numel())
                v = flat[off:off + n].view(shapes[f])
                setattr(self.desc.blk[i], "g_" + f, v.data_ptr())
                views.append(v)
                off += n
        if self.has_final_ln:
            for f in ("g_lnf_w", "g_lnf_b"):
                v = flat[off:off + self.D]
                setattr(self.desc, f, v.data_ptr())
                views.append(v)
                off += self.D
        self._keep["grads"] = flat
        return views

    def bind_grad_tensors(self, blocks_g: Sequence[Dict[str, torch.Tensor]], lnf_g):
        for i, bg in enumerate(blocks_g):
            for f in BLOCK_FIELDS:
                setattr(self.desc.blk[i], "g_" + f, bg[f].data_ptr())
        if self.has_final_ln:
            self.desc.g_lnf_w, self.desc.g_lnf_b = lnf_g[0].data_ptr(), lnf_g[1].data_ptr()
        self._keep["grads"] = (blocks_g, lnf_g)

    # ---- launches --------------------------------------------------------------------------------------
    def forward(self, x0: torch.Tensor, x0_ss: int, B: int, out: torch.Tensor, out_ss: int,
                pooled: Optional[torch.Tensor], training: bool, seed: int, step: int,
                step_dev: Optional[torch.Tensor] = None):
        if training:
            self.ensure_buffers(B)
        self.ensure_workspace(B)
        L.check(L.lib().m2m_tower_forward(C.byref(self.desc), x0.data_ptr(), x0_ss, B, out.data_ptr(), out_ss,
                                          L.ptr(pooled), int(training), seed & 0xFFFFFFFF, step & 0xFFFFFFFF,
                                          L.ptr(step_dev), L.stream_ptr()), "tower_forward")

    def backward(self, B: int, d_out: Optional[torch.Tensor], d_out_ss: int, d_pooled: Optional[torch.Tensor],
                 d_x0: torch.Tensor, d_x0_ss: int, seed: int, step: int, step_dev: Optional[torch.Tensor] = None):
        L.check(L.lib().m2m_tower_backward(C.byref(self.desc), B, L.ptr(d_out), d_out_ss, L.ptr(d_pooled),
                                           d_x0.data_ptr(), d_x0_ss, seed & 0xFFFFFFFF, step & 0xFFFFFFFF,
                                           L.ptr(step_dev), L.stream_ptr()), "tower_backward")

    def enable_dx0_image(self, B: int) -> bool:
        """Let the backward also leave d_x0^T as packed operand blocks (m2m_tower.dx0_chn): the patch-embedding weight
        gradient of the tower's embedding then takes the single-owner form (towers_wgrad(..., embed_towers=...)).  Fused-path
        bf16 towers only."""
        if self.wide or self.prec != L.PREC_BF16 or self.nblocks == 0:
            return False
        self._keep["want_dx0_image"] = True
        self._bufB = 0                       # (re)allocate with the image
        self.ensure_buffers(B)
        return True

    # ---- weight-gradient launch options (include/m2mixer.h: wgrad_flags, wslot) ----------------------------------------
    def wgrad_form(self, B: int) -> int:
        """1: the library's weight-gradient launch recomputes the hidden activation for this tower at batch B."""
        return int(L.lib().m2m_wgrad_form(C.byref(self.desc), B))

    def wgrad_groups(self, B: int) -> int:
        return int(L.lib().m2m_wgrad_groups(C.byref(self.desc), B))

    def channel_grad_ranges(self, flat_g: torch.Tensor):
        """[(lo, n)] per block: the flat-gradient index range of g_ch_w1 | g_ch_b1 | g_ch_w2 when the three lie back to back
        inside `flat_g` (the engines' layout), else None."""
        out = []
        cd = self.C * self.D
        base, end = flat_g.data_ptr(), flat_g.data_ptr() + flat_g.numel() * 4
        for i in range(self.nblocks):
            b = self.desc.blk[i]
            w1, b1, w2 = b.g_ch_w1, b.g_ch_b1, b.g_ch_w2
            if not w1 or b1 != w1 + 4 * cd or w2 != b1 + 4 * self.C or w1 < base or w2 + 4 * cd > end:
                return None
            out.append(((w1 - base) // 4, 2 * cd + self.C))
        return out

    def alloc_wslot(self, flat_g: torch.Tensor) -> bool:
        """Partial-gradient slot of every block (second row group of the weight-gradient launch), each with the 16-byte
        phase of its flat-gradient range so that the optimizer can add it with vector loads."""
        rng = self.channel_grad_ranges(flat_g)
        if rng is None or self.nblocks == 0:
            return False
        n = rng[0][1]
        buf = torch.zeros(self.nblocks * (n + 8), device=self.device)
        views = []
        off = 0
        for i, (lo, _) in enumerate(rng):
            while (off - lo) % 4:          # (buf is 256-byte aligned): element `off` gets the phase of flat element `lo`
                off += 1
            v = buf[off:off + n]
            self.desc.wslot[i] = v.data_ptr()
            views.append(v)
            off += n
        self._keep["wslot"] = (buf, views)
        return True

    def clear_wslot(self):
        for i in range(L.MAX_BLOCKS):
            self.desc.wslot[i] = None
        self._keep.pop("wslot", None)

    def wslot_views(self):
        return self._keep["wslot"][1] if "wslot" in self._keep else None

    def set_wgrad_overwrite(self, on: bool):
        self.desc.wgrad_flags = (self.desc.wgrad_flags | L.WGRAD_OVERWRITE) if on else (self.desc.wgrad_flags & ~L.WGRAD_OVERWRITE)

    def set_wgrad_reduces_small(self, on: bool):
        """Every backward of this tower is followed by a weight-gradient launch that includes it: the reduction of the backward's
        per-workgroup slots (small parameter gradients) rides in that launch (include/m2mixer.h: M2M_WGRAD_REDUCES_SMALL)."""
        self.desc.wgrad_flags = (self.desc.wgrad_flags | L.WGRAD_REDUCES_SMALL) if on else (self.desc.wgrad_flags & ~L.WGRAD_REDUCES_SMALL)

    def has_small_slots(self) -> bool:
        """The fused backward of this tower can collect its small parameter gradients in per-workgroup slots (ensure_split
        allocates them; csrc/tower_bwd.hip: m2m_small_part)."""
        return (not self.wide and self.prec == L.PREC_BF16 and self.D == 128 and self.nblocks > 0
                and 2 * self.T * self.N + self.T + self.N <= 576)

    def set_wgrad_group_slots(self, on: bool):
        """Small parameter gradients of the two-tower backward launch through per-workgroup slots (M2M_WGRAD_GROUP_SLOTS)."""
        self.desc.wgrad_flags = (self.desc.wgrad_flags | L.WGRAD_GROUP_SLOTS) if on else (self.desc.wgrad_flags & ~L.WGRAD_GROUP_SLOTS)

    def wgrad_fold(self):
        L.check(L.lib().m2m_wgrad_fold(C.byref(self.desc), L.stream_ptr()), "wgrad_fold")

    def backward_heads_ok(self, B: int, nheads: int, K: int) -> bool:
        return bool(L.lib().m2m_tower_backward_heads_ok(C.byref(self.desc), B, nheads, K))

    def backward_heads(self, B: int, heads: Sequence[dict], own: int, labels: torch.Tensor, K: int, out, d_x0: torch.Tensor,
                       d_x0_ss: int, seed: int, step: int, step_dev: Optional[torch.Tensor] = None):
        """Backward of this tower with the classification heads + multi-head CE in the launch's prologue (include/m2mixer.h:
        m2m_tower_backward_heads).  heads: dicts as for heads_ce; out = (logits, losses, preds)."""
        arr = _head_array(heads)
        logits, losses, preds = out
        L.check(L.lib().m2m_tower_backward_heads(C.byref(self.desc), B, arr, len(heads), own, labels.data_ptr(), K, logits.data_ptr(),
                                                 losses.data_ptr(), preds.data_ptr(), d_x0.data_ptr(), d_x0_ss, seed & 0xFFFFFFFF,
                                                 step & 0xFFFFFFFF, L.ptr(step_dev), L.stream_ptr()), "tower_backward_heads")

    def wgrad(self, B: int, seed: int, step: int, step_dev: Optional[torch.Tensor] = None):
        L.check(L.lib().m2m_tower_wgrad(C.byref(self.desc), B, seed & 0xFFFFFFFF, step & 0xFFFFFFFF, L.ptr(step_dev),
                                        L.stream_ptr()), "tower_wgrad")

    def device_desc(self) -> int:
        """Device pointer of a byte copy of the descriptor (for the multi-tower launches, whose kernel arguments cannot
        hold several 2.4 KiB descriptors).  Re-uploaded only when the descriptor changed -- never inside a graph capture:
        the warm-up steps before a capture leave it current."""
        raw = bytes(self.desc)
        if self._keep.get("desc_bytes") != raw:
            self._keep["desc_dev"] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
            self._keep["desc_bytes"] = raw
        return self._keep["desc_dev"].data_ptr()

    def dropout_mask(self, blk: int, site: int, B: int, seed: int, step: int) -> torch.Tensor:
        """Keep-mask (uint8) of one dropout site in the kernels' index order (test hook)."""
        n = {0: B * self.D * self.T, 1: B * self.D * self.N, 2: B * self.N * self.Cp, 3: B * self.N * self.D}[site]
        m = torch.empty(n, dtype=torch.uint8, device=self.device)
        L.check(L.lib().m2m_dropout_mask(C.byref(self.desc), blk, site, B, seed & 0xFFFFFFFF, step & 0xFFFFFFFF,
                                         m.data_ptr(), L.stream_ptr()), "dropout_mask")
        return m


def can_group(a: TowerRuntime, b: TowerRuntime, B: Optional[int] = None) -> bool:
    """Two towers may share their launches (m2m_towers_forward / _backward): fused path, same kernel instantiation -- or, with the
    batch size given, whatever the library takes as a pair at that batch (m2m_towers_can_group: also wide pairs at small batch)."""
    if B is not None:
        return bool(L.lib().m2m_towers_can_group(C.byref(a.desc), C.byref(b.desc), B))
    return (not a.wide and not b.wide and a.prec == b.prec and a.D == b.D and a.desc.p_drop == b.desc.p_drop
            and (a.N <= 4) == (b.N <= 4) and (a.T % 16 == 0) == (b.T % 16 == 0) and a.nblocks <= 4 and b.nblocks <= 4)


def towers_forward_embeds_ok(towers: Sequence[TowerRuntime], embeds: Sequence["EmbedRuntime"], B: int) -> bool:
    """towers_forward(..., embeds=, inputs=) takes this pair: the patch embeddings ride in the towers' launch."""
    n = len(towers)
    host = (C.POINTER(L.Tower) * n)(*[C.pointer(t.desc) for t in towers])
    ep = (C.POINTER(L.Embed) * n)(*[C.pointer(e.desc) for e in embeds])
    return bool(L.lib().m2m_towers_forward_embeds_ok(host, n, ep, B))


def towers_forward(towers: Sequence[TowerRuntime], ios: Sequence[tuple], B: int, training: bool, seed: int, step: int,
                   step_dev: Optional[torch.Tensor] = None, embeds: Sequence["EmbedRuntime"] = (), inputs: Sequence[torch.Tensor] = (),
                   head: Optional[tuple] = None):
    """ios[i] = (x0, x0_ss, out, out_ss, pooled or None[, x0_parts, x0_part_stride]): two towers, one launch.  With
    x0_parts > 1 the input is the sum of that many buffers (the k-split partial sums of embeds_forward).
    embeds + inputs (one per tower): the launch computes the patch embeddings itself (m2m_towers_forward_embeds; ios[i][0] is
    the (B N, D) scratch); head = (adam_state, losses): Adam step count += 1 and losses = 0 at the head of the launch."""
    n = len(towers)
    for t in towers:
        if training:
            t.ensure_buffers(B)
        t.ensure_workspace(B)
    host = (C.POINTER(L.Tower) * n)(*[C.pointer(t.desc) for t in towers])
    io = (L.TowerIO * n)()
    for i, (x0, x0_ss, out, out_ss, pooled, *parts) in enumerate(ios):
        io[i].x0, io[i].x0_ss, io[i].out, io[i].out_ss, io[i].pooled = x0.data_ptr(), x0_ss, out.data_ptr(), out_ss, L.ptr(pooled)
        io[i].x0_parts, io[i].x0_part_stride = parts if parts else (1, 0)
    if len(embeds):
        ep = (C.POINTER(L.Embed) * n)(*[C.pointer(e.desc) for e in embeds])
        ip = (C.c_void_p * n)(*[t.data_ptr() for t in inputs])
        hd = None
        if head is not None:
            hd = L.StepHead()
            hd.adam_state, hd.drop_counter, hd.losses, hd.nlosses = L.ptr(head[0]), None, L.ptr(head[1]), int(head[1].numel())
        L.check(L.lib().m2m_towers_forward_embeds(host, io, n, ep, ip, C.byref(hd) if hd is not None else None, B, int(training),
                                                  seed & 0xFFFFFFFF, step & 0xFFFFFFFF, L.ptr(step_dev), L.stream_ptr()),
                "towers_forward_embeds")
        return
    L.check(L.lib().m2m_towers_forward(host, io, n, B, int(training), seed & 0xFFFFFFFF, step & 0xFFFFFFFF, L.ptr(step_dev),
                                       L.stream_ptr()), "towers_forward")


def towers_backward(towers: Sequence[TowerRuntime], ios: Sequence[tuple], B: int, seed: int, step: int,
                    step_dev: Optional[torch.Tensor] = None):
    """ios[i] = (d_out or None, d_out_ss, d_pooled or None, d_x0, d_x0_ss): two towers, one launch."""
    n = len(towers)
    host = (C.POINTER(L.Tower) * n)(*[C.pointer(t.desc) for t in towers])
    io = (L.TowerGIO * n)()
    for i, (d_out, d_out_ss, d_pooled, d_x0, d_x0_ss) in enumerate(ios):
        io[i].d_out, io[i].d_out_ss, io[i].d_pooled = L.ptr(d_out), d_out_ss, L.ptr(d_pooled)
        io[i].d_x0, io[i].d_x0_ss = d_x0.data_ptr(), d_x0_ss
    L.check(L.lib().m2m_towers_backward(host, io, n, B, seed & 0xFFFFFFFF, step & 0xFFFFFFFF, L.ptr(step_dev), L.stream_ptr()),
            "towers_backward")


def wgrad_slot_groups(towers: Sequence[TowerRuntime], B: int) -> int:
    """bit i: towers_wgrad(towers, B) leaves tower i's second row group in its slot (TowerRuntime.alloc_wslot)."""
    n = len(towers)
    host = (C.POINTER(L.Tower) * n)(*[C.pointer(t.desc) for t in towers])
    return int(L.lib().m2m_wgrad_slot_groups(host, n, B))


def towers_wgrad(towers: Sequence[TowerRuntime], B: int, embeds: Sequence["EmbedRuntime"] = (),
                 inputs: Sequence[torch.Tensor] = (), d_x0s: Sequence[torch.Tensor] = (), seed: int = 0, step: int = 0,
                 step_dev: Optional[torch.Tensor] = None, embed_towers: Sequence[TowerRuntime] = (),
                 heads: Optional[Sequence[dict]] = None, K: int = 0, bump: Optional[torch.Tensor] = None):
    """Channel-mixing weight gradients of several towers (same precision / hidden_dim) in one launch; with `embeds`
    (the model's two patch embeddings, their inputs and d_x0) also the embedding gradients, in the same launch.
    seed / step / step_dev: the dropout stream of the forward (the recompute form regenerates the hidden keep-mask).
    embed_towers[i]: the tower embedding i feeds; with its d_x0^T image (enable_dx0_image) the embedding gradients take the
    single-owner form (no atomics).
    heads (dicts as for heads_ce, each with "g_part") + K: the launch also adds the heads' per-workgroup weight-gradient sums."""
    n, ne = len(towers), len(embeds)
    host = (C.POINTER(L.Tower) * n)(*[C.pointer(t.desc) for t in towers])
    dev = (C.c_void_p * n)(*[t.device_desc() for t in towers])
    ep = (C.POINTER(L.Embed) * max(ne, 1))(*[C.pointer(e.desc) for e in embeds])
    ip = (C.c_void_p * max(ne, 1))(*[t.data_ptr() for t in inputs])
    dp = (C.c_void_p * max(ne, 1))(*[t.data_ptr() for t in d_x0s])
    et = (C.POINTER(L.Tower) * max(ne, 1))(*[C.pointer(t.desc) for t in embed_towers]) if len(embed_towers) == ne and ne else None
    if bump is not None:
        # the step's dropout counter advances in this launch, behind its last reader (m2m_towers_wgrad_tail)
        L.check(L.lib().m2m_towers_wgrad_tail(host, dev, n, ep, ip, dp, et, ne, B, seed & 0xFFFFFFFF, step & 0xFFFFFFFF,
                                              L.ptr(step_dev), _head_array(heads) if heads else None, len(heads) if heads else 0, K,
                                              bump.data_ptr(), L.stream_ptr()), "towers_wgrad_tail")
        return
    if heads:
        # the classification heads' weight-gradient slots (heads_ce with "g_part") are added to g_w / g_b by this launch
        L.check(L.lib().m2m_towers_wgrad_heads(host, dev, n, ep, ip, dp, et, ne, B, seed & 0xFFFFFFFF, step & 0xFFFFFFFF,
                                               L.ptr(step_dev), _head_array(heads), len(heads), K, L.stream_ptr()), "towers_wgrad_heads")
        return
    L.check(L.lib().m2m_towers_wgrad(host, dev, n, ep, ip, dp, et, ne, B, seed & 0xFFFFFFFF, step & 0xFFFFFFFF, L.ptr(step_dev),
                                     L.stream_ptr()), "towers_wgrad")


def can_pack_all(towers: Sequence[TowerRuntime], embeds: Sequence["EmbedRuntime"]) -> bool:
    precs = {t.prec for t in towers} | {e.prec for e in embeds}
    return len(towers) <= 3 and len(embeds) <= 2 and len(precs) == 1 and all(t.nblocks <= 4 for t in towers)


def pack_all(towers: Sequence[TowerRuntime], embeds: Sequence["EmbedRuntime"]):
    """Every packed operand copy of a model in one launch (after the optimizer step)."""
    nt, ne = len(towers), len(embeds)
    tp = (C.POINTER(L.Tower) * max(nt, 1))(*[C.pointer(t.desc) for t in towers])
    ep = (C.POINTER(L.Embed) * max(ne, 1))(*[C.pointer(e.desc) for e in embeds])
    L.check(L.lib().m2m_pack_all(tp, nt, ep, ne, L.stream_ptr()), "pack_all")
    for t in towers:
        t.mark_packed()
    for e in embeds:
        e.mark_packed()


class AdamPackPlan:
    """Host + device copies of the plan of m2m_adam_pack_all (Adam and the operand re-pack of a whole model in one launch)."""

    def __init__(self, towers: Sequence[TowerRuntime], embeds: Sequence["EmbedRuntime"], flat_p, flat_g, grad_bf16, flat_m, flat_v,
                 state, betas, eps: float, weight_decay: float, grad_scale: float, ranges=None):
        """ranges: [(lo, n, slot tensor or None, keep), ...] -- the special gradient ranges of m2m_adam_step_ranges."""
        self.towers, self.embeds = list(towers), list(embeds)
        nt, ne = len(towers), len(embeds)
        self._tp = (C.POINTER(L.Tower) * max(nt, 1))(*[C.pointer(t.desc) for t in towers])
        self._ep = (C.POINTER(L.Embed) * max(ne, 1))(*[C.pointer(e.desc) for e in embeds])
        nbytes = int(L.lib().m2m_adam_pack_plan_bytes())
        self.host = C.create_string_buffer(nbytes)
        ranges = list(ranges or [])
        arr = (L.GradRange * max(len(ranges), 1))()
        for i, (rlo, rn, add, keep) in enumerate(ranges):
            arr[i].lo, arr[i].n, arr[i].add, arr[i].keep = rlo, rn, L.ptr(add), int(keep)
        L.check(L.lib().m2m_adam_pack_plan_ranges(self._tp, nt, self._ep, ne, flat_p.data_ptr(), flat_g.data_ptr(), L.ptr(grad_bf16),
                                                  flat_m.data_ptr(), flat_v.data_ptr(), flat_p.numel(), state.data_ptr(), betas[0], betas[1],
                                                  eps, weight_decay, abs(grad_scale), arr, len(ranges), self.host), "adam_pack_plan_ranges")
        self.dev = torch.frombuffer(bytearray(self.host.raw), dtype=torch.uint8).to(flat_p.device)
        self._keep = (flat_p, flat_g, grad_bf16, flat_m, flat_v, state, [r[2] for r in ranges])

    def run(self):
        L.check(L.lib().m2m_adam_pack_all(self._tp, len(self.towers), self._ep, len(self.embeds), self.dev.data_ptr(), self.host,
                                          L.stream_ptr()), "adam_pack_all")
        for t in self.towers:
            t.mark_packed()
        for e in self.embeds:
            e.mark_packed()


class EmbedRuntime:
    """m2m_embed: Conv2d(Cin, D, (ph, pw), stride=(ph, pw)) + rearrange, or Linear(K, D) on (B, N, K) rows."""

    def __init__(self, Cin: int, H: int, W: int, ph: int, pw: int, D: int, prec: int):
        self.desc = L.Embed()
        d = self.desc
        kb = 32 if prec == L.PREC_BF16 else 16
        K = Cin * ph * pw
        d.prec, d.Cin, d.H, d.W, d.ph, d.pw, d.D, d.K = prec, Cin, H, W, ph, pw, D, K
        d.Kp = (K + kb - 1) // kb * kb
        self.N = (H // ph) * (W // pw)
        self.D, self.K, self.prec = D, K, prec
        self._keep = {}
        self._packed_for = None

    def bind_params(self, w: torch.Tensor, b: torch.Tensor):
        if not w.is_cuda:
            raise RuntimeError("patch embedding: m2_mixer_amd runs on the GPU only; there is no CPU path")
        if w.dtype != torch.float32 or not w.is_contiguous() or w.numel() != self.D * self.K:
            raise RuntimeError(f"embedding weight: expected contiguous float32 with {self.D * self.K} elements")
        self.desc.w, self.desc.b = w.data_ptr(), b.data_ptr()
        self._keep["w"], self._keep["b"] = w, b
        self._packed_for = None            # new storage: the packed copy belongs to the old one
        if "wn" not in self._keep:
            wn = torch.zeros(L.packed_bytes(self.prec, self.D, self.desc.Kp), dtype=torch.uint8, device=w.device)
            self.desc.wn = wn.data_ptr()
            self._keep["wn"] = wn

    def params_changed(self, w, b) -> bool:
        return w.data_ptr() != self.desc.w or b.data_ptr() != self.desc.b

    def mark_packed(self):
        self._packed_for = self._keep["w"]._version

    def pack(self, force: bool = False):
        if force or self._keep["w"]._version != self._packed_for:
            L.check(L.lib().m2m_pack_embed(C.byref(self.desc), L.stream_ptr()), "pack_embed")
            self.mark_packed()

    def bind_grads(self, g_w: torch.Tensor, g_b: torch.Tensor):
        self.desc.g_w, self.desc.g_b = g_w.data_ptr(), g_b.data_ptr()
        self._keep["g"] = (g_w, g_b)

    def set_wgrad_overwrite(self, on: bool):
        """g_w is WRITTEN by the single-owner weight-gradient form (towers_wgrad with embed_towers), not accumulated."""
        self.desc.wgrad_flags = L.WGRAD_OVERWRITE if on else 0

    def fwd_splits(self) -> int:
        """k-splits embeds_forward should use for this embedding (1: none)."""
        return int(L.lib().m2m_embed_fwd_splits(C.byref(self.desc)))

    def forward(self, inp: torch.Tensor, B: int, x0: torch.Tensor, step_head: Optional[tuple] = None):
        """step_head = (adam_state, drop_counter, losses): the launch also does the step prologue (m2m_embed_forward_head)."""
        head = None
        if step_head is not None:
            adam_state, drop_counter, losses = step_head
            head = C.byref(L.StepHead(adam_state.data_ptr(), drop_counter.data_ptr(), losses.data_ptr(), losses.numel()))
        L.check(L.lib().m2m_embed_forward_head(C.byref(self.desc), inp.data_ptr(), B, x0.data_ptr(), head, L.stream_ptr()),
                "embed_forward")

    def wgrad(self, inp: torch.Tensor, d_x0: torch.Tensor, B: int):
        L.check(L.lib().m2m_embed_wgrad(C.byref(self.desc), inp.data_ptr(), d_x0.data_ptr(), B, L.stream_ptr()),
                "embed_wgrad")


def can_group_embeds(a: EmbedRuntime, b: EmbedRuntime) -> bool:
    return a.prec == b.prec and a.D == b.D


def embeds_forward(embeds: Sequence[EmbedRuntime], inputs: Sequence[torch.Tensor], x0s: Sequence[torch.Tensor], B: int,
                   nsplits: Optional[Sequence[int]] = None, step_head: Optional[tuple] = None):
    """Both patch embeddings of a two-tower model in one launch.  nsplits[i] > 1: x0s[i] is (nsplits[i], B*N, D) and
    receives k-split partial sums (EmbedRuntime.fwd_splits says when that pays off); the consumer adds them.
    step_head = (adam_state, drop_counter, losses): the launch also does the step prologue (m2m_step_prologue)."""
    n = len(embeds)
    ep = (C.POINTER(L.Embed) * n)(*[C.pointer(e.desc) for e in embeds])
    ip = (C.c_void_p * n)(*[t.data_ptr() for t in inputs])
    xp = (C.c_void_p * n)(*[t.data_ptr() for t in x0s])
    ns = (C.c_int * n)(*(nsplits if nsplits is not None else [1] * n))
    ps = (C.c_int64 * n)(*[B * e.N * e.D for e in embeds])
    head = None
    if step_head is not None:
        adam_state, drop_counter, losses = step_head
        head = C.byref(L.StepHead(adam_state.data_ptr(), drop_counter.data_ptr(), losses.data_ptr(), losses.numel()))
    L.check(L.lib().m2m_embeds_forward(ep, ip, xp, ns, ps, n, B, head, L.stream_ptr()), "embeds_forward")


def embeds_wgrad(embeds: Sequence[EmbedRuntime], inputs: Sequence[torch.Tensor], d_x0s: Sequence[torch.Tensor], B: int):
    """Weight / bias gradients of both patch embeddings in one launch."""
    n = len(embeds)
    ep = (C.POINTER(L.Embed) * n)(*[C.pointer(e.desc) for e in embeds])
    ip = (C.c_void_p * n)(*[t.data_ptr() for t in inputs])
    dp = (C.c_void_p * n)(*[t.data_ptr() for t in d_x0s])
    L.check(L.lib().m2m_embeds_wgrad(ep, ip, dp, n, B, L.stream_ptr()), "embeds_wgrad")


class MlpRuntime:
    """m2m_mlp: num_blocks x (Linear -> ReLU -> Dropout) + output Linear (modules/mlp.py:4-27)."""

    def __init__(self, dims: Sequence[int], has_out: bool, p_drop: float, site_base: int = 0):
        nl = len(dims) - 1
        if nl < 1 or nl > L.MLP_MAX_LAYERS:
            raise RuntimeError(f"MLP: 1..{L.MLP_MAX_LAYERS} Linear layers supported, got {nl}")
        self.dims, self.nlayers, self.has_out = list(dims), nl, bool(has_out)
        self.desc = L.Mlp()
        d = self.desc
        d.nlayers, d.has_out, d.p_drop, d.site_base = nl, int(self.has_out), float(p_drop), site_base
        for i, w in enumerate(dims):
            d.dims[i] = int(w)
        self._keep: Dict[str, object] = {}
        self._B = 0

    def bind(self, params: Sequence[tuple], grads: Optional[Sequence[tuple]], B: int):
        """params / grads: [(weight, bias)] per Linear, weight (dims[i+1], dims[i])."""
        for i, (w, b) in enumerate(params):
            _check_tensor(w, (self.dims[i + 1], self.dims[i]), f"mlp layer {i} weight")
            _check_tensor(b, (self.dims[i + 1],), f"mlp layer {i} bias")
            self.desc.w[i], self.desc.b[i] = w.data_ptr(), b.data_ptr()
        if grads is not None:
            for i, (gw, gb) in enumerate(grads):
                self.desc.g_w[i], self.desc.g_b[i] = gw.data_ptr(), gb.data_ptr()
        self._keep["params"], self._keep["grads"] = list(params), grads
        self.ensure_buffers(B, params[0][0].device)

    def ensure_buffers(self, B: int, device):
        if B == self._B:
            return
        self.use_acts([torch.zeros(B, self.dims[i + 1], device=device) for i in range(self.nlayers)])
        self._B = B

    def fresh_acts(self, B: int, device) -> list:
        """A new set of saved hidden activations for one forward (re-entrancy of the module path, see
        TowerRuntime.fresh_saved)."""
        acts = [torch.zeros(B, self.dims[i + 1], device=device) for i in range(self.nlayers)]
        self.use_acts(acts)
        self._B = B
        return acts

    def use_acts(self, acts: list):
        for i, a in enumerate(acts):
            self.desc.act[i] = a.data_ptr()
        self._keep["act"] = acts

    def forward(self, x: torch.Tensor, B: int, out: torch.Tensor, out_ss: int, out_dense: Optional[torch.Tensor],
                training: bool, seed: int, step: int, step_dev: Optional[torch.Tensor] = None):
        _check_tensor(x, (B, self.dims[0]), "mlp input")
        self.ensure_buffers(B, x.device)
        L.check(L.lib().m2m_mlp_forward(C.byref(self.desc), x.data_ptr(), B, out.data_ptr(), out_ss, L.ptr(out_dense),
                                        int(training), seed & 0xFFFFFFFF, step & 0xFFFFFFFF, L.ptr(step_dev),
                                        L.stream_ptr()), "mlp_forward")

    def backward(self, x: torch.Tensor, B: int, d_out: Optional[torch.Tensor], d_out_ss: int,
                 d_out_dense: Optional[torch.Tensor]):
        L.check(L.lib().m2m_mlp_backward(C.byref(self.desc), x.data_ptr(), B, L.ptr(d_out), d_out_ss, L.ptr(d_out_dense),
                                         L.stream_ptr()), "mlp_backward")

    # The same two calls RECORDED (m2m_mlp_forward_ride / _backward_ride): the next wide tower's forward / backward on this thread
    # carries the MLP's workgroups in its first token-mixing launch; ride_flush() launches a recorded call nothing carried.
    def forward_ride(self, x: torch.Tensor, B: int, out: torch.Tensor, out_ss: int, out_dense: Optional[torch.Tensor],
                     training: bool, seed: int, step: int, step_dev: Optional[torch.Tensor] = None):
        _check_tensor(x, (B, self.dims[0]), "mlp input")
        self.ensure_buffers(B, x.device)
        L.check(L.lib().m2m_mlp_forward_ride(C.byref(self.desc), x.data_ptr(), B, out.data_ptr(), out_ss, L.ptr(out_dense),
                                             int(training), seed & 0xFFFFFFFF, step & 0xFFFFFFFF, L.ptr(step_dev)), "mlp_forward_ride")

    def backward_ride(self, x: torch.Tensor, B: int, d_out: Optional[torch.Tensor], d_out_ss: int,
                      d_out_dense: Optional[torch.Tensor]):
        L.check(L.lib().m2m_mlp_backward_ride(C.byref(self.desc), x.data_ptr(), B, L.ptr(d_out), d_out_ss, L.ptr(d_out_dense)),
                "mlp_backward_ride")

    @staticmethod
    def ride_flush():
        L.check(L.lib().m2m_mlp_ride_flush(L.stream_ptr()), "mlp_ride_flush")


def _head_array(heads: Sequence[dict]):
    nh = len(heads)
    arr = (L.Head * nh)()
    for i, h in enumerate(heads):
        arr[i].pooled, arr[i].w, arr[i].b = L.ptr(h.get("pooled")), h["w"].data_ptr(), h["b"].data_ptr()
        tok = h.get("tokens")                               # (tensor or data pointer, ntok, sample stride in floats): the head pools itself
        if tok is not None:
            arr[i].tokens = tok[0] if isinstance(tok[0], int) else tok[0].data_ptr()
            arr[i].ntok, arr[i].tok_sample_stride = int(tok[1]), int(tok[2])
        elif h.get("pooled") is None:
            raise ValueError("a head needs `pooled` or `tokens`")
        arr[i].g_w, arr[i].g_b, arr[i].d_pooled = L.ptr(h.get("g_w")), L.ptr(h.get("g_b")), L.ptr(h.get("d_pooled"))
        arr[i].weight = float(h["weight"])
        arr[i].g_part = L.ptr(h.get("g_part"))
    return arr


def heads_bce(heads: Sequence[dict], targets: torch.Tensor, pos_weight: torch.Tensor, B: int, D: int, K: int, out=None,
              zero_losses: bool = True):
    """BCEWithLogitsLoss(pos_weight) heads (models/mmimdb.py:47-50): targets (B, K) float32 multi-hot.
    Returns logits (nh, B, K), losses (nh + 1), preds (nh, B, K) int32."""
    nh = len(heads)
    arr = _head_array(heads)
    _check_tensor(targets, (B, K), "BCE targets")
    dev = targets.device
    if out is not None:
        logits, losses, preds = out
    else:
        logits = torch.empty(nh, B, K, device=dev)
        losses = torch.empty(nh + 1, device=dev)
        preds = torch.empty(nh, B, K, dtype=torch.int32, device=dev)
    L.check(L.lib().m2m_heads_bce(arr, nh, targets.data_ptr(), pos_weight.data_ptr(), B, D, K, logits.data_ptr(),
                                  losses.data_ptr(), preds.data_ptr(), int(zero_losses), L.stream_ptr()), "heads_bce")
    return logits, losses, preds


def heads_ce(heads: Sequence[dict], labels: torch.Tensor, B: int, D: int, K: int, out=None, zero_losses: bool = True):
    """heads: dicts with pooled, w, b, g_w, g_b, d_pooled (tensors or None) and weight.
    Returns logits (nh, B, K), losses (nh + 1), preds (nh, B) int32 (written into `out` if given)."""
    nh = len(heads)
    arr = _head_array(heads)
    dev = labels.device
    if out is not None:
        logits, losses, preds = out
    else:
        logits = torch.empty(nh, B, K, device=dev)
        losses = torch.empty(nh + 1, device=dev)
        preds = torch.empty(nh, B, dtype=torch.int32, device=dev)
    L.check(L.lib().m2m_heads_ce(arr, nh, labels.data_ptr(), B, D, K, logits.data_ptr(), losses.data_ptr(),
                                 preds.data_ptr(), int(zero_losses), L.stream_ptr()), "heads_ce")
    return logits, losses, preds
