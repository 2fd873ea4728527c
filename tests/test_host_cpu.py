"""CPU-side checks (run with -m "not gpu"): the C-ABI library loads and exports every declared symbol,
the reference-shaped host surface (registry, ctor signatures, state-dict keys, fusion shape algebra),
loud failure on CPU tensors, the data-parallel gradient sync over gloo, bench.py's FLOP accounting."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import gen_util as G
from golden_util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from m2_mixer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "m2mixer.h")).read()
    declared = set(re.findall(r"\b(m2m_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found in include/m2mixer.h"
    L = lib.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, f"declared but not exported: {missing}"
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    assert L.m2m_abi_version() == lib.ABI_VERSION
    assert lib.packed_bytes(lib.PREC_BF16, 3104, 128) == L.m2m_packed_bytes(lib.PREC_BF16, 3104, 128)
    assert lib.packed_bytes(lib.PREC_F32, 3078, 128) == L.m2m_packed_bytes(lib.PREC_F32, 3078, 128)


def test_struct_layouts_match_header(lib):
    import ctypes as C
    # m2m_block: 12 params + 5 packed + 12 grads + 6 saved pointers; m2m_tower header is 40 bytes then 7 pointers
    assert C.sizeof(lib.Block) == 35 * 8
    assert C.sizeof(lib.Tower) == 40 + 7 * 8 + lib.MAX_BLOCKS * C.sizeof(lib.Block) + 8 + 8 + 8 + 8 + 3 * 8 * lib.MAX_BLOCKS + 8
    assert lib.Tower.slabs.offset == 96 + lib.MAX_BLOCKS * C.sizeof(lib.Block)
    assert lib.Tower.blk.offset == 96
    assert C.sizeof(lib.Embed) == 40 + 5 * 8 + 8
    assert C.sizeof(lib.Head) == 6 * 8 + 8 + 8 + 8 + 8 + 8 and lib.Head.g_part.offset == 56       # ABI 17: + tokens, tok_sample_stride, ntok
    assert lib.Head.tokens.offset == 64 and lib.Head.tok_sample_stride.offset == 72 and lib.Head.ntok.offset == 80


def test_struct_layouts_against_the_c_compiler(lib, tmp_path):
    """sizeof / offsetof of every struct that crosses the boundary, as gcc sees include/m2mixer.h, against the ctypes mirror."""
    import ctypes as C
    import subprocess
    fields = {"m2m_block": ("Block", ["ln1_w", "w1n", "g_ln1_w", "x_in", "dh_chn"]),
              "m2m_tower": ("Tower", ["p_drop", "lnf_w", "blk", "slabs", "wgrad_flags", "xres", "gpart", "a_nat", "dy_nat", "wslot", "dx0_chn"]),
              "m2m_embed": ("Embed", ["Kp", "w", "g_b", "wgrad_flags"]),
              "m2m_head": ("Head", ["d_pooled", "weight"]),
              "m2m_mlp": ("Mlp", ["dims", "p_drop", "w", "act"]),
              "m2m_grad_range": ("GradRange", ["lo", "n", "add", "keep"]),
              "m2m_step_head": ("StepHead", ["losses", "nlosses"])}
    # ctypes names that differ from the C field names
    alias = {("Tower", "x0_ss"): "x0_sample_stride"}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "m2mixer.h"', 'int main(void) {']
    for cname, (_, fl) in fields.items():
        src.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fl:
            src.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    src += ['printf("M2M_MAX_GRAD_RANGES %d\\n", M2M_MAX_GRAD_RANGES);', 'printf("M2M_WGRAD_OVERWRITE %d\\n", M2M_WGRAD_OVERWRITE);',
            'printf("M2M_ABI_VERSION %d\\n", M2M_ABI_VERSION);', 'return 0; }']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, (pyname, fl) in fields.items():
        cls = getattr(lib, pyname)
        assert int(got[cname]) == C.sizeof(cls), cname
        for f in fl:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"
    assert int(got["M2M_MAX_GRAD_RANGES"]) == lib.MAX_GRAD_RANGES
    assert int(got["M2M_WGRAD_OVERWRITE"]) == lib.WGRAD_OVERWRITE
    assert int(got["M2M_ABI_VERSION"]) == lib.ABI_VERSION


def test_registry_and_state_dict_keys():
    from m2_mixer_amd import modules as M
    cfg = G.AVMNIST["B"]
    img = M.get_block_by_name(block_type="MLPMixer", **cfg["image"], dropout=0.5, extra_key_ignored=1)
    aud = M.get_block_by_name(block_type="MLPMixer", **cfg["audio"], dropout=0.5)
    fusion = M.get_fusion_by_name(fusion_function="ConcatFusion", block_type="FusionMixer", hidden_dim=128)
    npatch = fusion.get_output_shape(img.num_patch, aud.num_patch, dim=1)
    assert (img.num_patch, aud.num_patch, npatch) == (4, 4, 8)
    fus = M.get_block_by_name(block_type="FusionMixer", fusion_function="ConcatFusion", **cfg["multimodal"],
                              num_patches=npatch, dropout=0.5)
    cls = M.get_classifier_by_name(classifier="StandardClassifier", num_classes=10, input_shape=[16, 49, 128],
                                   hidden_dims=[1024, 512, 256, 32])
    model = torch.nn.Module()
    model.image_mixer, model.audio_mixer, model.fusion_mixer = img, aud, fus
    model.classifier_image = torch.nn.Linear(128, 10)
    model.classifier_audio = torch.nn.Linear(128, 10)
    model.classifier_fusion = cls
    sd = model.state_dict()
    shapes = G.avmnist_shapes(cfg)          # verified against the reference's own state_dict in make_golden.py
    assert list(sd.keys()) == list(shapes.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(shapes[k]), k
    assert sum(v.numel() for v in sd.values()) == int(load("avmnist_B.npz")["n_params"]) == 8339354
    with pytest.raises(AttributeError):
        M.get_block_by_name(block_type="NoSuchBlock")
    with pytest.raises(AssertionError):
        M.MLPMixer(1, 32, 5, [28, 28], 1, 16, 64)


def test_no_patching_and_mlp_keys():
    from m2_mixer_amd import modules as M
    cfg = G.MIMIC_H
    t = M.get_block_by_name(block_type="MLPMixerNoPatching", in_channels=1, **cfg["time"], dropout=0.3)
    st = M.get_block_by_name(block_type="MLP", in_channels=1, **cfg["static"], dropout=0.3)
    m = torch.nn.Module()
    m.time_mixer, m.static_extractor = t, st
    want = [k for k in G.mimic_shapes(cfg) if k.startswith(("time_mixer.", "static_extractor."))]
    assert list(m.state_dict().keys()) == want
    # like every module here the MLP runs in the HIP library only (csrc/mlp.hip): a CPU tensor is refused
    with pytest.raises(RuntimeError, match="GPU only"):
        st(torch.randn(7, 5))


def test_cpu_tensors_fail_loudly():
    from m2_mixer_amd import modules as M
    blk = M.MixerBlock(32, 4, 16, 64)
    with pytest.raises(RuntimeError, match="GPU only"):
        blk(torch.zeros(2, 4, 32))
    mix = M.MLPMixer(1, 32, 14, [28, 28], 1, 16, 64)
    with pytest.raises(RuntimeError, match="GPU only"):
        mix(torch.zeros(2, 1, 28, 28))
    from m2_mixer_amd.engine import AVMnistEngine
    with pytest.raises(RuntimeError, match="GPU only"):
        AVMnistEngine(G.AVMNIST["S"], 4, device="cpu")


def test_fusion_shape_algebra_matches_reference():
    """The reference's tests/modules/test_fusion.py cases (the six that pass on the reference), checked
    against values captured from the reference (tests/golden/fusion_shapes.npz)."""
    from m2_mixer_amd.modules import ConcatFusion, SumFusion, MaxFusion, MeanFusion, ConcatDynaFusion, BiModalGatedUnit
    gold = load("fusion_shapes.npz")
    a, b = torch.rand(10, 20, 30), torch.rand(10, 20, 30)
    f = ConcatFusion(useless_arg=1)
    assert tuple(f(a, b).shape) == tuple(gold["concat//call"]) == (10, 40, 30)
    assert f.get_output_shape(a.shape, b.shape) == tuple(gold["concat//shape"])
    assert f.get_output_shape(20, 20, dim=1) == int(gold["concat//dim1"]) == 40
    assert f.get_output_shape(20, 20, dim=0) == int(gold["concat//dim0"]) == 20
    with pytest.raises(ValueError):
        f.get_output_shape(a, b, dim=2)
    for cls in (SumFusion, MaxFusion, MeanFusion):
        g = cls(useless_arg=1)
        assert tuple(g(a, b).shape) == tuple(gold[f"{cls.__name__}//call"])
        assert g.get_output_shape(a.shape, b.shape) == tuple(gold[f"{cls.__name__}//shape"])
        assert g.get_output_shape(20, 20, dim=1) == int(gold[f"{cls.__name__}//dim1"])
        assert g.get_output_shape(20, 20, dim=0) == 20
        with pytest.raises(ValueError):
            g.get_output_shape(a, b, dim=2)
    d = ConcatDynaFusion(useless_arg=1)
    a4, b4 = torch.rand(10, 20, 20, 30), torch.rand(10, 20, 20, 30)
    assert tuple(d(a4, b4).shape) == tuple(gold["dyna//call"])
    assert d.get_output_shape(a4.shape, b4.shape) == tuple(gold["dyna//shape"])
    assert d.get_output_shape(36, 36, dim=1) == int(gold["dyna//dim1"]) == 144
    with pytest.raises(ValueError):
        d.get_output_shape(a4, b4, dim=2)
    bi = BiModalGatedUnit(30, 30, 30, useless_arg=1)
    assert bi(a, b).shape == (10, 20, 30)
    assert bi.get_output_shape(a.shape, b.shape) == (10, 20, 30)
    assert bi.get_output_shape(20, 20, dim=1) == 20 and bi.get_output_shape(20, 20, dim=-1) == 30


def test_engine_param_layout_matches_reference_order():
    from m2_mixer_amd.engine import avmnist_param_shapes
    for size in ("S", "M", "B"):
        mine = avmnist_param_shapes(G.AVMNIST[size])
        ref = G.avmnist_shapes(G.AVMNIST[size])
        assert list(mine.items()) == [(k, tuple(v)) for k, v in ref.items()]


def test_bench_flop_accounting():
    sys.path.insert(0, ROOT)
    import bench
    # SURVEY.md section 8a / BASELINE.md: 235.846 MFLOP per training sample (79.753 forward) for M2-Mixer-B, 4.218 for S
    assert abs(bench.total_train_flops(bench.CFG_B, 512) / 512 / 1e6 - 235.846) < 0.01
    assert abs(bench.total_train_flops(bench.CFG_S, 32) / 32 / 1e6 - 4.218) < 0.01
    a = bench.algorithmic_flops(bench.CFG_B, 1)
    fwd = sum(sum(v.values()) for k, v in a.items() if k != "heads") + a["heads"]
    assert abs(fwd / 1e6 - 79.753) < 0.01


def test_erf_rational_coefficients():
    """The device erf (csrc/common.h erf_fast) restated in numpy float32 against scipy."""
    from scipy.special import erf
    src = open(os.path.join(ROOT, "m2_mixer_amd", "csrc", "common.h")).read()
    body = src[src.index("static __device__ __forceinline__ float erf_fast"):]
    body = body[:body.index("return p * __builtin_amdgcn_rcpf(q);")]
    coef = [np.float32(c) for c in re.findall(r"(-?\d\.\d+e-\d+)f", body)]
    assert len(coef) == 12
    x = np.linspace(-6, 6, 200001).astype(np.float32)
    xc = np.clip(x, -4, 4)
    x2 = xc * xc
    p = coef[0]
    for c in coef[1:7]:
        p = x2 * p + c
    p = xc * p
    q = coef[7]
    for c in coef[8:]:
        q = x2 * q + c
    assert np.abs(p / q - erf(x.astype(np.float64))).max() < 6e-7


def _gloo_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from m2_mixer_amd import parallel
    r, lr, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    p = torch.full((10,), float(rank))
    parallel.broadcast_parameters(p)
    scale = parallel.GradSync()(flat)
    scale16 = parallel.GradSync(compress="bf16")(flat16 := (torch.ones(64) * (rank + 1)))
    keep = parallel.GradSync(compress="bf16", widen=False)        # the sum stays in bf16 for an optimizer that reads it there
    local = torch.ones(64) * (rank + 1)
    keep(local)
    assert torch.equal(local, torch.ones(64) * (rank + 1)), "widen=False must leave the local gradient alone"
    assert torch.equal(keep.reduced_bf16.float(), torch.full((64,), 3.0))
    assert parallel.GradSync(compress="bf16").reduced_bf16 is None and parallel.GradSync().reduced_bf16 is None
    tmax = parallel.max_over_ranks(1.0 + rank, device="cpu")
    # plain lists, not tensors: a tensor crosses a torch.multiprocessing queue as a shared-memory handle that the parent
    # must open while this process still lives -- a worker that exits first makes q.get() raise EOFError
    q.put((rank, flat.tolist(), scale, p.tolist(), flat16.tolist(), scale16, tmax, parallel.shard_batch_seed(1234, rank)))
    torch.distributed.destroy_process_group()


def test_grad_sync_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, flat, scale, prm, flat16, scale16, tmax, seed in out:
        assert scale == 0.5 and scale16 == 0.5
        assert flat == (torch.arange(1000, dtype=torch.float32) * 3).tolist()      # sum over ranks
        assert prm == [0.0] * 10                                                     # broadcast from rank 0
        assert flat16 == [3.0] * 64
        assert tmax == 2.0
        assert seed == 1234 + rank


def _pipelined_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from m2_mixer_amd import parallel
    parallel.init_from_env(backend="gloo")
    gen = torch.Generator().manual_seed(100 + rank)
    g = torch.randn(5000, generator=gen)
    ref = g.clone()
    torch.distributed.all_reduce(ref)                              # what GradSync would leave in the buffer
    sync = parallel.PipelinedGradSync()
    bounds = [(0, 1700), (1700, 1701), (1701, 5000)]               # uneven chunks, one of a single element
    order = []
    scale = sync.start(g, bounds)
    for k, (lo, hi) in enumerate(bounds):                          # the consumer's loop: chunk k is complete after wait(k)
        sync.wait(k)
        order.append(bool(torch.equal(g[lo:hi], ref[lo:hi])))
    bad = []
    for b in ([(0, 10), (20, 5000)], [(0, 5000), (5000, 5000)], [(0, 4000)]):      # gap / empty chunk / not covering
        try:
            parallel.PipelinedGradSync().start(torch.zeros(5000), b)
            bad.append(b)
        except ValueError:
            pass
    whole = torch.randn(64, generator=gen)
    wref = whole.clone()
    torch.distributed.all_reduce(wref)
    s2 = parallel.PipelinedGradSync()(whole)                       # drop-in for GradSync: one chunk
    q.put((rank, scale, order, bool(torch.equal(g, ref)), bad, s2, bool(torch.equal(whole, wref))))
    torch.distributed.destroy_process_group()


def test_pipelined_grad_sync_gloo_world2():
    """parallel.PipelinedGradSync over gloo, two ranks: the chunks partition the buffer, chunk k holds the all-reduced values
    once wait(k) returns (the optimizer's consumption order), the whole buffer equals ONE all-reduce bit for bit, malformed
    chunk lists are refused, and called like a GradSync it reduces the buffer as one chunk."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, scale, order, same, bad, s2, same2 in out:
        assert scale == 0.5 and s2 == 0.5
        assert order == [True, True, True] and same and same2
        assert bad == []


def test_bench_cli_contract_help():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun environment must start two rank processes itself (before touching the
    GPU), relay rank 0's single JSON line and fail when a rank fails -- here over gloo, rendezvous only."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["M2M_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "world_size": 2, "max_rank": 1.0}
    # a torchrun-style environment whose world size disagrees with --gpus is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                         text=True, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_task_module_surface_and_no_cpu_path():
    """models.py mirrors the reference's task modules: built from cfg dicts through the registry, state-dict keys in the
    reference's creation order (checkpoint compatibility), optimizer as configured at models/avmnist.py:413-422 -- and,
    like every module here, no CPU compute path."""
    import torch
    from m2_mixer_amd import models as MD
    for task, cls, shapes_fn, c in (("avmnist", MD.AVMnistMixerMultiLoss, G.avmnist_shapes, G.AVMNIST["S"]),
                                    ("mimic", MD.MimicMixerMultiLoss, G.mimic_shapes, G.MIMIC_H),
                                    ("mmimdb", MD.MMIMDBMixerMultiLoss, G.mmimdb_shapes, G.MMIMDB)):
        fusion = dict(c["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion")
        if task == "mimic":
            mods = {"static": dict(c["static"], block_type="MLP"), "time": dict(c["time"], block_type="MLPMixerNoPatching")}
        else:
            a, b = ("image", "audio") if task == "avmnist" else ("image", "text")
            mods = {a: dict(c[a], block_type="MLPMixer"), b: dict(c[b], block_type="MLPMixer")}
        mods["multimodal"] = fusion
        mods["classification"] = dict(classifier="StandardClassifier", num_classes=c["num_classes"],
                                      input_shape=[16, 49, c["multimodal"]["hidden_dim"]])
        cfg = {"dropout": c["dropout"], "modalities": mods}
        if task == "mmimdb":
            cfg["pos_weight"] = c["pos_weight"]
        net = cls(cfg, {"lr": 1e-2, "scheduler_patience": 2})
        shapes = shapes_fn(c)
        sd = net.state_dict()
        extra = ["image_criterion.pos_weight", "text_criterion.pos_weight", "fusion_criterion.pos_weight"] if task == "mmimdb" else []
        assert list(sd.keys()) == list(shapes.keys()) + extra, task       # parameters, then the loss modules' buffers
        assert [k for k, _ in net.named_parameters()] == list(shapes.keys()), task
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes), task
        opt = net.configure_optimizers()
        assert isinstance(opt["optimizer"], torch.optim.Adam) and opt["optimizer"].defaults["lr"] == 1e-2
        assert opt["lr_scheduler"].patience == 2 and opt["monitor"] == "val_loss"
        with pytest.raises(NotImplementedError):
            cls(dict(cfg, use_softadapt=True), {"lr": 1e-2})
    net = MD.MimicMixerMultiLoss(cfg if task == "mimic" else
                                 {"dropout": 0.0, "modalities": {"static": dict(G.MIMIC_H["static"], block_type="MLP"),
                                                                 "time": dict(G.MIMIC_H["time"], block_type="MLPMixerNoPatching"),
                                                                 "multimodal": dict(G.MIMIC_H["multimodal"], block_type="FusionMixer",
                                                                                    fusion_function="ConcatFusion"),
                                                                 "classification": dict(classifier="StandardClassifier", num_classes=6,
                                                                                        input_shape=[16, 25, 64])}},
                                 {"lr": 1e-2})
    with pytest.raises(RuntimeError, match="GPU only"):
        net.shared_step(G.mimic_batch(2, 1, G.MIMIC_H))


def test_checkpoint_io_lightning_layout(tmp_path):
    """a Lightning-style `.ckpt` (top-level `state_dict` with the reference's key names, plus
    `hyper_parameters` pickling a class of a package that is not installed here) loads through load_from_checkpoint, which
    remembers the path; test_preds.pt gets the reference's keys (models/avmnist.py:382-398)."""
    import importlib
    import types
    import torch
    from m2_mixer_amd import models as MD
    c = G.AVMNIST["S"]
    mods = {"image": dict(c["image"], block_type="MLPMixer"), "audio": dict(c["audio"], block_type="MLPMixer"),
            "multimodal": dict(c["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion"),
            "classification": dict(classifier="StandardClassifier", num_classes=10, input_shape=[16, 49, c["multimodal"]["hidden_dim"]])}
    cfg, ocfg = {"dropout": 0.1, "modalities": mods}, {"lr": 1e-2, "scheduler_patience": 2}
    src = MD.AVMnistMixerMultiLoss(cfg, dict(ocfg))
    with torch.no_grad():
        for i, p in enumerate(src.parameters()):
            p.copy_(torch.randn_like(p) * 0.1 + i * 1e-3)
    # a foreign object inside the checkpoint whose class cannot be imported at load time
    fake = types.ModuleType("omegaconf_not_installed_here")
    fake.DictConfig = type("DictConfig", (), {"__module__": "omegaconf_not_installed_here", "__init__": lambda self, d=None: setattr(self, "d", d)})
    sys.modules["omegaconf_not_installed_here"] = fake
    ck = tmp_path / "version_0" / "checkpoints" / "epoch=3-step=400.ckpt"
    ck.parent.mkdir(parents=True)
    torch.save({"epoch": 3, "global_step": 400, "pytorch-lightning_version": "1.8.6", "state_dict": src.state_dict(),
                "hyper_parameters": {"model_cfg": fake.DictConfig({"x": 1})}, "optimizer_states": [{}], "lr_schedulers": [{}]}, ck)
    del sys.modules["omegaconf_not_installed_here"]
    with pytest.raises(Exception):
        torch.load(ck, weights_only=False)                       # the plain unpickler cannot resolve the foreign class
    # a file that needs the full unpickler is refused unless the caller vouches for it (unpickling runs code from the file)
    with pytest.raises(RuntimeError, match="trusted=True"):
        MD.AVMnistMixerMultiLoss.load_from_checkpoint(ck, optimizer_cfg=dict(ocfg), model_cfg=cfg)
    net = MD.AVMnistMixerMultiLoss.load_from_checkpoint(ck, optimizer_cfg=dict(ocfg), model_cfg=cfg, trusted=True)
    assert net.checkpoint_path == str(ck) and net.current_epoch == 3
    for (k, a), (k2, b) in zip(src.state_dict().items(), net.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    with pytest.raises(TypeError):
        MD.AVMnistMixerMultiLoss.load_from_checkpoint(ck)
    # our own checkpoints go through the same door -- with the safe loader (tensors and plain containers only)
    out = net.save_checkpoint(tmp_path / "own" / "last.ckpt", epoch=7, global_step=9)
    again = MD.AVMnistMixerMultiLoss.load_from_checkpoint(out, optimizer_cfg=dict(ocfg), model_cfg=cfg)
    assert again.current_epoch == 7 and all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), again.state_dict().values()))
    raw = torch.load(out, weights_only=True)
    from packaging.version import Version
    assert Version(raw["pytorch-lightning_version"]) == Version("1.8.6")      # Lightning's migration parses this field
    assert raw["optimizer_states"] == [] and raw["lr_schedulers"] == []
    # test_preds.pt next to the checkpoint, the reference's keys, batches concatenated
    outs = [{k: torch.full((4, 10) if "logits" in k else (4,), float(i)) for k in net.TEST_PRED_KEYS} for i in range(3)]
    path = net.save_test_preds(outs)
    assert os.path.dirname(path) == os.path.dirname(out) and os.path.basename(path) == "test_preds.pt"     # next to the last checkpoint
    dump = torch.load(path)
    assert sorted(dump) == sorted(["preds", "preds_image", "preds_audio", "labels", "image_logits", "audio_logits", "logits"])
    assert dump["logits"].shape == (12, 10) and dump["preds"].shape == (12,) and float(dump["labels"][-1]) == 2.0
    assert "preds_text" in MD.MMIMDBMixerMultiLoss.TEST_PRED_KEYS and MD.MimicMixerMultiLoss.TEST_PRED_KEYS == ()


def test_mmimdb_checkpoint_keys_round_trip_with_the_reference_layout(tmp_path):
    """The reference's MMIMDBMixerMultiLoss holds three nn.BCEWithLogitsLoss(pos_weight=...) submodules
    (models/mmimdb.py:47-50), so every reference state_dict ends with `image_criterion.pos_weight`,
    `text_criterion.pos_weight`, `fusion_criterion.pos_weight`.  Keys built here from the reference's construction order
    (the class itself needs pytorch_lightning to import): towers, fusion mixer, heads, then the three buffers."""
    import torch
    from m2_mixer_amd import models as MD
    c = G.MMIMDB
    mods = {"image": dict(c["image"], block_type="MLPMixer"), "text": dict(c["text"], block_type="MLPMixer"),
            "multimodal": dict(c["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion"),
            "classification": dict(classifier="StandardClassifier", num_classes=c["num_classes"],
                                   input_shape=[16, 49, c["multimodal"]["hidden_dim"]])}
    cfg, ocfg = {"dropout": 0.0, "modalities": mods, "pos_weight": c["pos_weight"]}, {"lr": 1e-3}
    ref_keys = list(G.mmimdb_shapes(c)) + ["image_criterion.pos_weight", "text_criterion.pos_weight", "fusion_criterion.pos_weight"]
    net = MD.MMIMDBMixerMultiLoss(cfg, dict(ocfg))
    assert list(net.state_dict().keys()) == ref_keys
    # a reference-written checkpoint: same keys, its own pos_weight values
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    pw = torch.arange(1, c["num_classes"] + 1, dtype=torch.float32)
    for k in ref_keys[-3:]:
        sd[k] = pw.clone()
    ck = tmp_path / "ref.ckpt"
    torch.save({"state_dict": sd, "epoch": 2, "global_step": 10, "pytorch-lightning_version": "1.8.6"}, ck)
    got = MD.MMIMDBMixerMultiLoss.load_from_checkpoint(ck, model_cfg=cfg, optimizer_cfg=dict(ocfg))      # strict=True
    assert torch.equal(got.fusion_criterion.pos_weight, pw) and got._engine_cfg()["pos_weight"] == pw.tolist()
    # and back: what save_checkpoint writes has exactly the reference's key list (a strict reference-side load succeeds)
    out = got.save_checkpoint(tmp_path / "own.ckpt")
    assert list(torch.load(out, weights_only=True)["state_dict"].keys()) == ref_keys


def _write_avmnist(root, n_train, n_test, seed=0, learnable=False):
    """A tiny dataset in the reference's on-disk format (datasets/avmnist.py:105-114)."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "image"), exist_ok=True)
    os.makedirs(os.path.join(root, "audio"), exist_ok=True)
    for stage, n in (("train", n_train), ("test", n_test)):
        labels = rng.integers(0, 10, size=n)
        image = rng.random((n, 784), dtype=np.float32)
        audio = rng.random((n, 112, 112), dtype=np.float32)
        if learnable:                         # class k brightens image rows / audio rows of band k
            for i, k in enumerate(labels):
                image[i].reshape(28, 28)[2 * k:2 * k + 3, :] += 2.0
                audio[i, 11 * k:11 * k + 11, :] += 1.0
        np.save(os.path.join(root, "image", f"{stage}_data.npy"), image)
        np.save(os.path.join(root, "audio", f"{stage}_data.npy"), audio)
        np.save(os.path.join(root, f"{stage}_labels.npy"), labels)


def test_resident_dataset_format_splits_and_sharding(tmp_path):
    from m2_mixer_amd.data import PlateauLR, ResidentAVMnist
    root = str(tmp_path / "avmnist")
    _write_avmnist(root, 240, 30)
    ds = ResidentAVMnist(root, device="cpu")
    image, audio, labels = ds.splits["train"]
    assert tuple(image.shape) == (220, 1, 28, 28) and tuple(audio.shape) == (220, 1, 112, 112) and labels.dtype == torch.int64
    assert ds.splits["val"][2].shape[0] == 20 and ds.splits["test"][2].shape[0] == 30        # 11 : 1 as 55 000 : 5 000
    raw = np.load(os.path.join(root, "image", "train_data.npy"))
    assert np.array_equal(image[:, 0].reshape(220, 784).numpy(), raw[:220])
    # every sample is served: the reference's DataLoaders keep the ragged last batch (drop_last=False, datasets/avmnist.py:180-190)
    got = list(ds.batches("train", 50))
    assert len(got) == 5 == ds.num_batches("train", 50) and [b[2].shape[0] for b in got] == [50, 50, 50, 50, 20]
    assert torch.equal(torch.cat([b[2] for b in got]), labels)
    assert len(list(ds.batches("train", 50, drop_last=True))) == 4 == ds.num_batches("train", 50, drop_last=True)
    assert got[1][0].data_ptr() == image[50:100].data_ptr(), "unshuffled single-rank batches must be views"
    assert [b[2].shape[0] for b in ds.batches("val", 8)] == [8, 8, 4]
    # two ranks: disjoint strided shards that together cover the split, as DistributedSampler(shuffle=False) hands them out
    r0 = torch.cat([b[2] for b in ResidentAVMnist(root, "cpu", 0, 2).batches("train", 50)])
    r1 = torch.cat([b[2] for b in ResidentAVMnist(root, "cpu", 1, 2).batches("train", 50)])
    assert torch.equal(r0, labels[0::2]) and torch.equal(r1, labels[1::2])
    three = [torch.cat([b[2] for b in ResidentAVMnist(root, "cpu", r, 3).batches("test", 4)]) for r in range(3)]
    assert [t.shape[0] for t in three] == [10, 10, 10] and sorted(torch.cat(three).tolist()) == sorted(ds.splits["test"][2].tolist())
    # shuffled test split: a permutation
    t = torch.cat([b[2] for b in ds.batches("test", 10, shuffle=True, generator=torch.Generator().manual_seed(1))])
    assert sorted(t.tolist()) == sorted(ds.splits["test"][2].tolist())

    class FakeEngine:
        lr = None

        def set_lr(self, lr):
            self.lr = lr

    eng = FakeEngine()
    sch = PlateauLR(eng, 1e-2, patience=2)
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-2)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, patience=2)
    for v in (1.0, 0.9, 0.95, 0.93, 0.92, 0.91, 0.5, 0.6, 0.6, 0.6, 0.6):
        ref.step(v)
        assert abs(sch.step(v) - ref_opt.param_groups[0]["lr"]) < 1e-12, v
    assert eng.lr is not None and eng.lr < 1e-2


# ---- data-parallel epoch loop -------------------------------------------------------
def _write_fake_avmnist(root, n, n_test=0):
    rng = np.random.default_rng(5)
    for sub in ("image", "audio"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    np.save(os.path.join(root, "image", "train_data.npy"), rng.random((n, 784), dtype=np.float32))
    np.save(os.path.join(root, "audio", "train_data.npy"), rng.random((n, 112, 112), dtype=np.float32))
    np.save(os.path.join(root, "train_labels.npy"), rng.integers(0, 10, size=(n,)))


def test_resident_data_matches_distributed_sampler(tmp_path):
    """Per-rank sample lists == torch's DistributedSampler(shuffle=False, drop_last=False), which Lightning puts in front of
    the reference's loaders under DDP: padded with the head of the index list, rank r takes r, r + world, ..."""
    import torch
    from torch.utils.data.distributed import DistributedSampler
    from m2_mixer_amd.data import ResidentAVMnist
    n = 131                                                       # train split: 131 * 11 // 12 = 120 ... plus odd remainders
    _write_fake_avmnist(str(tmp_path), n)
    for world in (1, 2, 3, 8):
        counts = set()
        for rank in range(world):
            data = ResidentAVMnist(str(tmp_path), device="cpu", rank=rank, world=world)
            ntrain = data.splits["train"][2].shape[0]
            want = list(DistributedSampler(range(ntrain), num_replicas=world, rank=rank, shuffle=False, drop_last=False))
            got = []
            for image, audio, labels in data.batches("train", 16):
                # recover the indices from the labels + first pixel (unique enough: compare the tensors themselves)
                got.append((image, labels))
            img_all = torch.cat([g[0] for g in got])
            ref = data.splits["train"][0][torch.tensor(want)]
            assert torch.equal(img_all, ref), (world, rank)
            assert data.num_samples("train") == len(want)
            counts.add((data.num_samples("train"), data.num_batches("train", 16)))
        assert len(counts) == 1, f"ranks disagree on their sample / batch counts: {counts}"


class _StubEngine:
    """The surface run_epoch drives, on the CPU: a linear 'model' whose train_step counts the gradient exchanges."""

    MODS = ("image", "audio")

    def __init__(self, batch_size, w=None, counter=None):
        import torch
        self.B = batch_size
        self.device = torch.device("cpu")
        self.flat_p = w if w is not None else torch.zeros(4)
        self.losses = torch.zeros(4)
        self.preds = torch.zeros(3, batch_size, dtype=torch.int32)
        self.exchanges = counter if counter is not None else [0]
        self.packs = 0

    def sibling(self, bs, trains=True):
        return _StubEngine(bs, self.flat_p, self.exchanges)

    def pack(self):
        self.packs += 1

    def train_step(self, image, audio, labels, grad_sync=None):
        import torch
        assert image.shape[0] == self.B
        g = torch.full((4,), float(image.mean()))
        if grad_sync is not None:
            scale = grad_sync(g)
            self.exchanges[0] += 1
        else:
            scale = 1.0
        self.flat_p -= 0.1 * scale * g
        self.losses = torch.full((4,), float(image.mean()))
        self.preds = torch.zeros(3, self.B, dtype=torch.int32)
        return self.losses

    def evaluate(self, image, audio, labels):
        return {}


def _epoch_worker(rank, world, port, root, q):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from m2_mixer_amd import parallel
    from m2_mixer_amd.data import ResidentAVMnist, run_epoch
    parallel.init_from_env(backend="gloo")
    data = ResidentAVMnist(root, device="cpu", rank=rank, world=world)
    eng = _StubEngine(16)
    sync = parallel.GradSync()
    try:
        run_epoch(eng, data, "train", 16, train=True)       # world > 1 without an exchange must be refused
        refused = False
    except RuntimeError:
        refused = True
    out = run_epoch(eng, data, "train", 16, train=True, grad_sync=sync)
    q.put((rank, eng.exchanges[0], out["steps"], out["samples"], eng.flat_p.tolist(), refused))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_run_epoch_world2_same_collectives_and_parameters(tmp_path):
    """An odd sample count (121 training samples, 2 ranks, batch 16): both ranks see 61 samples = 3 full batches + a tail of
    13, exchange gradients in EVERY step (the tail included) and end the epoch with identical parameters."""
    import torch.multiprocessing as mp
    _write_fake_avmnist(str(tmp_path), 132)                       # -> 121 training samples
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, ex0, st0, n0, p0, ref0), (r1, ex1, st1, n1, p1, ref1) = out
    assert ref0 and ref1, "run_epoch(world > 1, train) without grad_sync must raise"
    assert (st0, n0) == (st1, n1) == (4, 61)
    assert ex0 == ex1 == 4, "every step -- the ragged last one too -- exchanges gradients"
    assert p0 == p1, "parameters diverged across ranks"


def test_bf16_compressed_exchange_error_bound():
    """`--grad-compress bf16` rounds every rank's gradient to bf16 and lets the collective add in bf16.  Bound of an 8-way
    sum against the fp32 exchange (DDP's semantics, bench.py's default), worst order (a sequential ring: seven roundings on the
    running sum): elementwise |err| <= 8 * 2^-8 * sum_r |g_r| (each of the 8 input roundings and 7 partial-sum roundings is
    at most half a bf16 ulp = 2^-9 relative, of a value no larger than the sum of magnitudes), and on gradients shaped like the
    model's (heavy-tailed, rank-to-rank correlation 0.5) the relative L2 error stays below 1 %."""
    import torch
    g = torch.Generator().manual_seed(0)
    n, world = 1 << 16, 8
    common = torch.randn(n, generator=g)
    scale = torch.exp(2.0 * torch.randn(n, generator=g)) * 1e-3           # log-normal magnitudes over ~4 decades
    grads = [(0.7 * common + 0.7 * torch.randn(n, generator=g)) * scale for _ in range(world)]
    exact = torch.stack(grads).double().sum(0)
    acc = grads[0].to(torch.bfloat16)
    for r in range(1, world):
        acc = (acc.float() + grads[r].to(torch.bfloat16).float()).to(torch.bfloat16)      # the collective's add, in bf16
    err = (acc.double() - exact).abs()
    bound = 8 * 2.0 ** -8 * torch.stack(grads).double().abs().sum(0)
    assert bool((err <= bound + 1e-30).all())
    rel_l2 = float(err.norm() / exact.norm())
    assert rel_l2 < 1e-2, rel_l2
    print(f"bf16 8-way exchange: relative L2 error {rel_l2:.2e}, max elementwise error / bound {float((err / (bound + 1e-30)).max()):.2f}")
