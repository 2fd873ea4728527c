#!/usr/bin/env python3
"""Headline benchmark: AV-MNIST M2-Mixer-B training samples/s (fwd + bwd + Adam), bf16, per-GPU batch 512.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit/..., plus
  "roofline"     dominant kernel: algorithmic FLOPs per launch / its mean duration (HIP events on the launch
                 stream) against the dense bf16 MFMA peak,
  "cpu_baseline" the CPU oracle's training step timed on this box's host cores (rank 0, N = 1 only); the other CPU points of
                 BASELINE.md section 3 in "cpu_baselines_other_configs"; "eager_rocm_baseline": the same eager ops on this GPU.
Synthetic data of the dataset's shape (image U[0,1) (B,1,28,28), audio U[0,1) (B,1,112,112), labels randint(10),
numpy default_rng(1234 + rank)), torch-default random init under seed 42.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense peaks, MI355X_MICROARCH.md "Chip-level parameters"
# HBM bytes per launch come from the rocprofv3 PMC passes (counters cannot be collected from inside this process): the
# summary scripts/collect_profiles.sh writes, keyed by kernel name and by the hash of the kernel sources it was taken on.
PMC_JSON = os.path.join(ROOT, "profiles", "r04_pmc.json")
LAUNCH_KERNEL = {"towers_bwd[image+audio]": "tower_bwd_group_kernel", "tower_bwd[fusion]": "tower_bwd_kernel",
                 "tower_bwd[fusion]+heads": "tower_bwd_heads_kernel",
                 "towers_fwd[image+audio]": "tower_fwd_group_kernel", "tower_fwd[fusion]": "tower_fwd_kernel",
                 "towers_wgrad[all+embeds]": "tower_wgrad_group_kernel", "adam+pack": "adam_pack_all_kernel",
                 "embeds_fwd[image+audio]": "embed_fwd_group_kernel", "heads_ce": "heads_kernel"}


def pmc_traffic(launch, model, B, precision):
    """HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) of one launch of `launch` from profiles/r04_pmc.json -- None, with a
    warning, when the file is missing, was taken on other kernel sources, or covers another configuration."""
    if not (model == "B" and B == 512 and precision == "bf16"):
        return None
    try:
        with open(PMC_JSON) as f:
            rec = json.load(f)
    except OSError:
        log(f"roofline.traffic: {PMC_JSON} not found -> null")
        return None
    from m2_mixer_amd import _lib
    if rec.get("csrc_sha256") != _lib.csrc_hash():
        log("roofline.traffic: profiles/r04_pmc.json was collected on different kernel sources (stale) -> null; "
            "re-run scripts/collect_profiles.sh")
        return None
    names = LAUNCH_KERNEL.get(launch, "")
    names = names if isinstance(names, tuple) else (names,)       # a span of several launches (Adam, then the re-pack): their sum
    ks = [rec["kernels"].get(n) for n in names]
    return None if any(k is None for k in ks) else sum(k["traffic_bytes"] for k in ks)


# AV-MNIST M2-Mixer-B  (reference cfg/avmnist/avmnist_m2-mixer_B.yml:24-56)
CFG_B = dict(dropout=0.5, num_classes=10,
             image=dict(in_channels=1, hidden_dim=128, patch_size=14, image_size=[28, 28], token_dim=32, channel_dim=3072, num_mixers=4),
             audio=dict(in_channels=1, hidden_dim=128, patch_size=56, image_size=[112, 112], token_dim=32, channel_dim=3072, num_mixers=4),
             multimodal=dict(hidden_dim=128, token_dim=32, channel_dim=3078, num_mixers=2))
CFG_S = dict(dropout=0.1, num_classes=10,
             image=dict(in_channels=1, hidden_dim=32, patch_size=14, image_size=[28, 28], token_dim=16, channel_dim=256, num_mixers=2),
             audio=dict(in_channels=1, hidden_dim=32, patch_size=56, image_size=[112, 112], token_dim=16, channel_dim=256, num_mixers=2),
             multimodal=dict(hidden_dim=32, token_dim=16, channel_dim=256, num_mixers=1))


def n_patch(c):
    return (c["image_size"][0] // c["patch_size"]) * (c["image_size"][1] // c["patch_size"])


def algorithmic_flops(cfg, B):
    """2*MACs of every Linear/Conv: forward, dgrad, wgrad (no recompute, no elementwise) -- SURVEY.md section 8d.
    Returns per-launch-kind forward FLOPs; dgrad == wgrad == forward for every GEMM."""
    out = {}
    for name, c, N in (("image", cfg["image"], n_patch(cfg["image"])), ("audio", cfg["audio"], n_patch(cfg["audio"])),
                       ("fusion", cfg["multimodal"], n_patch(cfg["image"]) + n_patch(cfg["audio"]))):
        D, T, C, nb = c["hidden_dim"], c["token_dim"], c["channel_dim"], c["num_mixers"]
        M = B * N
        chan = 2 * 2 * M * D * C * nb
        tok = 2 * 2 * B * D * N * T * nb
        out[name] = {"channel": chan, "token": tok}
        if "patch_size" in c:
            out[name]["embed"] = 2 * M * D * c["in_channels"] * c["patch_size"] ** 2
    K = cfg["num_classes"]
    out["heads"] = 3 * 2 * B * K * cfg["image"]["hidden_dim"]
    return out


def total_train_flops(cfg, B):
    a = algorithmic_flops(cfg, B)
    fwd = sum(sum(v.values()) for k, v in a.items() if k != "heads") + a["heads"]
    # input gradients of the two patch embeddings are not needed (dgrad of embed skipped): fwd + dgrad + wgrad
    emb = sum(v.get("embed", 0) for k, v in a.items() if k != "heads")
    return 3 * fwd - emb


def make_batch(cfg, B, seed, device):
    rng = np.random.default_rng(seed)
    ih, iw = cfg["image"]["image_size"]
    ah, aw = cfg["audio"]["image_size"]
    image = torch.from_numpy(rng.random((B, 1, ih, iw), dtype=np.float32)).to(device)
    audio = torch.from_numpy(rng.random((B, 1, ah, aw), dtype=np.float32)).to(device)
    labels = torch.from_numpy(rng.integers(0, cfg["num_classes"], size=(B,), dtype=np.int64)).to(device)
    return image, audio, labels


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_util as G
    from oracle import m2mixer_oracle as O
    return G, O


def host_cores():
    # the GPU box gives one job a 16-core share whatever os.cpu_count() says: oversubscribing it is pathological
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, int(os.environ.get("M2M_CPU_THREADS", "16"))))


def baseline_point(name, cfg, B, device, threads=None, budget_s=10.0, autocast=None):
    """The oracle's training step (fwd + autograd bwd + Adam, dropout masks drawn like nn.Dropout) -- the restatement of
    the reference's eager PyTorch path -- timed on `device`: the CPU (cpu_baseline, `threads` host threads) or, as
    BASELINE config 2's "vs eager" comparator, the same eager ops on the GPU through torch-ROCm.  Bounded to ~budget_s."""
    G, O = _oracle()
    on_gpu = torch.device(device).type == "cuda"
    if not on_gpu:
        torch.set_num_threads(threads)
    params = {k: v.to(device) for k, v in G.make_params(G.avmnist_shapes(cfg), 42).items()}
    image, audio, labels = make_batch(cfg, B, 1234, device)
    state, p = {}, cfg["dropout"]
    gen = None if on_gpu else torch.Generator().manual_seed(0)

    def step():
        masks = O.avmnist_random_masks(cfg, B, p, gen, device=device) if p > 0 else None
        if autocast is not None:
            with torch.autocast("cuda", dtype=autocast):
                O.avmnist_train_step(image, audio, labels, params, cfg, state, lr=1e-2, drop_p=p, masks=masks)
        else:
            O.avmnist_train_step(image, audio, labels, params, cfg, state, lr=1e-2, drop_p=p, masks=masks)
        if on_gpu:
            torch.cuda.synchronize()

    t0 = time.perf_counter()
    step()                                    # warm-up (allocator, thread pool, kernel selection)
    if on_gpu:
        step()
        t0 = time.perf_counter()
        step()
    warm = time.perf_counter() - t0
    n = max(2, min(50, int(budget_s / max(warm, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    flops = total_train_flops(cfg, B)
    what = (f"eager torch-ROCm ({'fp32' if autocast is None else 'bf16 autocast'})" if on_gpu
            else f"fp32 CPU oracle (oracle/m2mixer_oracle.py), {threads} thread(s)")
    return {"value": round(B * n / dt, 2), "unit": "samples/s", "cores": None if on_gpu else threads, "kind": "port",
            "config": name, "gflops": round(flops * n / dt / 1e9, 1),
            "sample": f"{n} training steps of the {what}, {name}, batch {B}, {dt / n * 1e3:.1f} ms/step, torch {torch.__version__}"}


def cpu_baselines(budget_s=20.0):
    """the north-star workload (M2-Mixer-B, batch 512) on all host cores -- the `cpu_baseline` of the
    bench line -- plus config 1 (M2-Mixer-S, fp32, batch 32) and M2-Mixer-B at batch 32, on all cores and on one thread."""
    cores = host_cores()
    main = baseline_point("AV-MNIST M2-Mixer-B", CFG_B, 512, "cpu", cores, 0.55 * budget_s)
    log(f"cpu baseline: {main['sample']}")
    extra = []
    for name, cfg, B, thr in (("AV-MNIST M2-Mixer-S (BASELINE config 1)", CFG_S, 32, cores), ("AV-MNIST M2-Mixer-S (BASELINE config 1)", CFG_S, 32, 1),
                              ("AV-MNIST M2-Mixer-B", CFG_B, 32, cores), ("AV-MNIST M2-Mixer-B", CFG_B, 32, 1)):
        extra.append(baseline_point(name, cfg, B, "cpu", thr, 0.1125 * budget_s))
    torch.set_num_threads(cores)
    return main, extra


def shader_clock_mhz(device, spin_ms=1.0):
    """Median shader clock over the chip's 256 CUs while every CU runs MFMA + VALU work for ~spin_ms (m2m_clock_probe:
    s_memtime cycles / s_memrealtime ticks x 100 MHz).  Recorded before and after the timed region: this launch-bound, partly
    issue-bound step runs at 1.8-2.1 GHz depending on the box and on what the kernel does, far below the 2.4 GHz maximum."""
    from m2_mixer_amd import _lib as L
    nwg = 256
    out = torch.zeros(2 * nwg, dtype=torch.int64, device=device)
    L.check(L.lib().m2m_clock_probe(out.data_ptr(), nwg, int(spin_ms * 1e5), L.stream_ptr()), "clock_probe")
    torch.cuda.synchronize()
    v = out.view(nwg, 2).cpu().numpy().astype(np.float64)
    ticks = np.bitwise_and(out.view(nwg, 2)[:, 1].cpu().numpy(), (1 << 62) - 1).astype(np.float64)
    return round(float(np.median(v[:, 0] / np.maximum(ticks, 1.0))) * 100.0, 1)


def module_path_point(cfg, B, device, precision, budget_s=4.0):
    """BASELINE config 2 for the IMPORT-SWAP path: what a user of the reference gets from replacing `modules` by
    `m2_mixer_amd.modules` and nothing else -- AVMnistMixerMultiLoss built from the cfg dicts through the registry
    (modules/__init__.py:12-26), shared_step (models/avmnist.py:236-312) -> loss.backward() under torch autograd ->
    torch.optim.Adam.step() (models/avmnist.py:413-415), eager launches, no graph, no fused optimizer.  Outside the timed region."""
    import m2_mixer_amd as M
    from m2_mixer_amd import models as MD
    M.set_precision(precision)
    mods = {"image": dict(cfg["image"], block_type="MLPMixer"), "audio": dict(cfg["audio"], block_type="MLPMixer"),
            "multimodal": dict(cfg["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion"),
            "classification": dict(classifier="StandardClassifier", num_classes=cfg["num_classes"],
                                   input_shape=[B, n_patch(cfg["image"]) + n_patch(cfg["audio"]), cfg["multimodal"]["hidden_dim"]])}
    torch.manual_seed(42)
    net = MD.AVMnistMixerMultiLoss({"dropout": cfg["dropout"], "modalities": mods}, {"lr": 1e-2, "betas": (0.9, 0.999), "scheduler_patience": 2}).to(device)
    net.train()
    opt = net.configure_optimizers()["optimizer"]
    image, audio, labels = make_batch(cfg, B, 1234, device)
    batch = {"image": image, "audio": audio, "label": labels}

    def step():
        opt.zero_grad(set_to_none=True)
        out = net.shared_step(batch, mode="train")
        out["loss"].backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    n = max(5, min(200, int(budget_s / max(time.perf_counter() - t0, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"value": round(B * n / dt, 1), "unit": "samples/s", "ms_per_step": round(dt / n * 1e3, 4), "steps": n,
           "what": "m2_mixer_amd.models.AVMnistMixerMultiLoss (registry-built m2_mixer_amd.modules towers under torch autograd): "
                   f"shared_step -> loss.backward() -> torch.optim.Adam.step(), eager, {precision}, batch {B}"}
    # the same step replayed as ONE hipGraph (m2_mixer_amd.graphs.GraphedStep: capturable Adam, dropout counters on the device)
    try:
        from m2_mixer_amd import config as MC
        from m2_mixer_amd.graphs import GraphedStep
        torch.manual_seed(42)
        net2 = MD.AVMnistMixerMultiLoss({"dropout": cfg["dropout"], "modalities": mods}, {"lr": 1e-2, "betas": (0.9, 0.999), "scheduler_patience": 2}).to(device)
        net2.train()
        gs = GraphedStep(net2, net2.configure_optimizers()["optimizer"], batch)
        for _ in range(5):
            gs(batch)
        torch.cuda.synchronize()
        n2 = 200
        t0 = time.perf_counter()
        for _ in range(n2):
            gs(batch)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        out["graphed"] = {"value": round(B * n2 / dt2, 1), "unit": "samples/s", "ms_per_step": round(dt2 / n2 * 1e3, 4), "steps": n2,
                          "what": "the same shared_step -> backward -> Adam(capturable) step replayed as one hipGraph (m2_mixer_amd.graphs.GraphedStep)"}
        MC.set_device_dropout_step(False)
    except Exception as e:                                   # (a diagnostic leg: never takes the bench line down)
        out["graphed"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    return out


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def launch_ranks(n: int, argv, script=None) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes (one per GPU) through
    torch.distributed.run, BEFORE this process has made any GPU call, relay rank 0's JSON line and the ranks' stderr, and
    return non-zero if any rank failed.  (The driver may equally start the ranks itself; then WORLD_SIZE is set and this
    is skipped.)"""
    import socket
    import subprocess
    with socket.socket() as sk:                                   # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    log(f"--gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd[1:8])} ...")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            lines.append(line.strip())
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc != 0:
        log(f"a rank failed (torch.distributed.run exit code {rc})")
        return rc
    if len(lines) != 1:
        log(f"expected ONE JSON line from rank 0, got {len(lines)}")
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (cfg batch_size is per process)")
    ap.add_argument("--model", default="B", choices=["B", "S"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--steps-per-graph", type=int, default=10,
                    help="N = 1: training steps captured per hipGraph, EACH WITH ITS OWN INPUT SLOT (ten distinct synthetic batches, "
                         "270 MB: a step reads its batch from HBM, as a training epoch does).  The one-step graph replays just as "
                         "densely and measures 0.3-0.8 %% faster (scripts/spg_probe.sh) -- because it re-reads ONE 27 MB batch that "
                         "stays in the 256 MiB Infinity Cache, which no real epoch does: not the default.  Remainder steps run "
                         "through a single-step graph")
    ap.add_argument("--grad-compress", default="none", choices=["none", "bf16"],
                    help="N > 1: dtype of the gradient all-reduce.  none (default) = fp32: what the reference's DDP exchanges "
                         "(run.py:69-70).  bf16 = the equivalent of DDP's bf16_compress_hook, half the bytes on xGMI; its error is "
                         "bounded in tests/test_host_cpu.py::test_bf16_compressed_exchange_error_bound")
    ap.add_argument("--grad-exchange", default="pipelined", choices=["pipelined", "single"],
                    help="N > 1, fp32 exchange: pipelined (default) = one all-reduce per parameter segment on a communication stream, "
                         "Adam + re-pack of segment k behind the all-reduce of segment k + 1 (parallel.PipelinedGradSync); single = ONE "
                         "all-reduce of the flat gradient, then Adam + re-pack (parallel.GradSync).  Same arithmetic either way")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="~this many ms of untimed steps (preheat_ms / 0.75 of them) before the W warm-up steps, so a fresh box's "
                         "clocks have ramped up when the warm-up starts (reported in config.preheat_ms; 0 disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-module-path", action="store_true", help="skip the import-swap module-path leg (outside the timed region)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--profile-steps", type=int, default=10, help="eager steps with HIP events around every launch")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins the process group, the ranks agree on a MAX, rank 0 prints a JSON line "
                         "(exercises the --gpus N launcher without a GPU: M2M_DIST_BACKEND=gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))             # nothing has touched the GPU yet in this process

    from m2_mixer_amd import parallel
    from m2_mixer_amd.engine import AVMnistEngine

    rank, local_rank, world = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the job must run exactly one rank per requested GPU")
    if world > 1:
        assert torch.distributed.get_world_size() == args.gpus
    if args.launch_check:
        top = parallel.max_over_ranks(float(rank), "cpu" if torch.distributed.is_initialized() and torch.distributed.get_backend() == "gloo" else None)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "world_size": world, "max_rank": top}), flush=True)
        return
    dev = torch.device(f"cuda:{int(os.environ.get('M2M_FORCE_DEVICE', local_rank))}")      # override: single-GPU rehearsal of N > 1
    torch.cuda.set_device(dev)
    cfg = CFG_B if args.model == "B" else CFG_S
    B = args.batch

    log(f"rank {rank}/{world}: building engine (model {args.model}, batch {B}, {args.precision})")
    eng = AVMnistEngine(cfg, B, device=dev, precision=args.precision, lr=1e-2, seed=42)
    parallel.broadcast_parameters(eng.flat_p)
    eng.pack()
    image, audio, labels = make_batch(cfg, B, parallel.shard_batch_seed(1234, rank), dev)
    compress = None if args.grad_compress == "none" or args.precision == "fp32" else args.grad_compress
    exchange = None
    if world > 1:
        if compress is None and args.grad_exchange == "pipelined":
            sync, exchange = parallel.PipelinedGradSync(), "pipelined"
        else:
            sync, exchange = parallel.GradSync(compress=compress, widen=False), "single"      # (bf16: Adam reads the bf16 sum directly)
    else:
        sync = None

    spg = 1 if (args.no_graph or world > 1) else max(1, args.steps_per_graph)
    if args.no_graph:
        def step():
            eng.train_step(image, audio, labels, grad_sync=sync)
        multi = None
    else:
        try:
            replay = eng.capture(image, audio, labels, grad_sync=sync)
        except Exception as e:                     # the pipelined exchange has only ever run over gloo (one-GPU boxes): if its first
            if exchange != "pipelined":            # contact with RCCL fails on EVERY rank alike (same code path), fall back loudly
                raise
            log(f"rank {rank}: pipelined exchange failed at capture ({type(e).__name__}: {str(e)[:200]}); falling back to the single all-reduce")
            sync, exchange = parallel.GradSync(), "single (fallback: the pipelined exchange failed at capture)"
            eng = AVMnistEngine(cfg, B, device=dev, precision=args.precision, lr=1e-2, seed=42)
            parallel.broadcast_parameters(eng.flat_p)
            eng.pack()
            replay = eng.capture(image, audio, labels, grad_sync=sync)

        def step():
            replay()
        multi = eng.capture(image, audio, labels, steps=spg) if spg > 1 else None

    def run_steps(n):
        """n training steps: as many multi-step graphs as fit, the remainder one step at a time."""
        if multi is not None:
            for _ in range(n // spg):
                multi()
            n = n % spg
        for _ in range(n):
            step()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The clock probe (~1 ms of back-to-back MFMA + VALU work per CU, then a host read-back) goes in FRONT of the preheat: between
    # the preheat and the timed region it left the chip in another power state for the next few milliseconds -- with the driver's
    # 20 timed steps (10 ms) that cost 4-8 % (K = 20: 0.516-0.535 ms per step against 0.495 at K = 200 on the same box; with the
    # probe moved the two agree).  Preheat, warm-up and the timed steps now run back to back, separated only by the barriers.
    clk = [shader_clock_mhz(dev)] if rank == 0 else []
    if args.preheat_ms > 0:
        n_pre = int(args.preheat_ms / 0.75)                 # a fixed step count: every rank must issue the same collectives
        log(f"preheat: {n_pre} untimed steps (~{args.preheat_ms:.0f} ms)")
        run_steps(n_pre)
    run_steps(args.warmup)
    barrier()
    log(f"timing {args.steps} steps")
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    if clk:
        clk.append(shader_clock_mhz(dev))
    # the last step ran through the multi-step graph (its own per-step output slots) or through the single-step graph
    last_multi = multi is not None and args.steps >= spg and args.steps % spg == 0
    loss_end = float((multi.losses[-1] if last_multi else eng.losses)[3])
    log(f"timed region {elapsed * 1e3:.1f} ms; profiling launches")

    # ---- per-launch timing (HIP events on the launch stream), eager pass of the very same step ----
    kern = profile_launches(eng, image, audio, labels, args.profile_steps) if rank == 0 else None

    def finish():
        """All ranks leave together: a peer that exits while rank 0 is still profiling would tear the communicator down
        under it."""
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()

    if rank != 0:
        finish()
        return
    ms = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    flops_step = total_train_flops(cfg, B)
    peak = MFMA_PEAK_TFLOPS[args.precision]
    dom = max(kern.items(), key=lambda kv: kv[1]["us_per_step"])
    roof = {"bound": "mfma", "kernel": dom[0], "achieved": round(dom[1]["flops_per_launch"] / (dom[1]["us_per_launch"] * 1e-6) / 1e12, 2),
            "peak": peak, "unit": "TFLOP/s",
            "traffic": pmc_traffic(dom[0], args.model, B, args.precision)}
    roof["frac"] = round(roof["achieved"] / peak, 4)
    out = {
        "metric": "training samples/sec AV-MNIST M2-Mixer-%s %s" % (args.model, args.precision),
        "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"AV-MNIST M2-Mixer-{args.model}: fwd + bwd + Adam, dropout {cfg['dropout']}, per-GPU batch {B}, "
                               f"global batch {B * world}, {eng.n_params} params",
                   "parallelism": f"dp{world}", "world_size": world, "launch": "eager" if args.no_graph else ("hipGraph" if spg == 1 else f"hipGraph, {spg} steps per graph"),
                   "grad_allreduce": (compress or "fp32") if world > 1 else None, "grad_exchange": exchange, "preheat_ms": args.preheat_ms},
        "roofline": roof,
        "step_mfma_frac": round(world * B * args.steps / elapsed / world * flops_step / B / (peak * 1e12), 4),
        "algorithmic_gflop_per_step": round(flops_step / 1e9, 2),
        "kernels_us": {k: round(v["us_per_step"], 1) for k, v in kern.items()},
        "final_loss": round(loss_end, 4),
        "shader_clock_mhz": {"before": clk[0], "after": clk[1], "max": 2400,
                             "how": "m2m_clock_probe: median over 256 CUs of s_memtime / s_memrealtime during ~1 ms of MFMA + VALU work"},
    }
    if world == 1 and not args.no_cpu_baseline:
        # baselines, all outside the timed region: the eager path on this GPU (BASELINE config 2's comparator), then the CPU
        log("eager torch-ROCm baseline (the oracle's ops on the GPU) ...")
        out["eager_rocm_baseline"] = [baseline_point(f"AV-MNIST M2-Mixer-{args.model}", cfg, B, dev, budget_s=3.0, autocast=ac)
                                      for ac in (None, torch.bfloat16)]
        if not args.no_module_path:
            log("module path (import-swap: m2_mixer_amd.modules under torch autograd + torch.optim.Adam) ...")
            out["module_path"] = module_path_point(cfg, B, dev, args.precision)
        log("cpu baselines (oracle) ...")
        out["cpu_baseline"], out["cpu_baselines_other_configs"] = cpu_baselines(args.cpu_budget)
    print(json.dumps(out), flush=True)
    finish()


def profile_launches(eng, image, audio, labels, nsteps):
    """Median duration of every launch of the training step -- the very launches the replayed graph holds (the step
    runs on ONE stream, so an eager pass is already serialized) -- measured with events recorded on the stream the
    launches go to (torch's current stream == the stream handed to libm2mixer)."""
    cfg, B = eng.cfg, eng.B
    alg = algorithmic_flops(cfg, B)
    spans = {}

    def timed(name, fn, flops):
        def wrapper(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            spans.setdefault(name, {"events": [], "flops": flops})["events"].append((e0, e1))
            return r
        return wrapper

    import m2_mixer_amd.engine as E
    from m2_mixer_amd import _lib as L
    two = ("image", "audio")
    f_tow = lambda t: alg[t]["channel"] + alg[t]["token"]
    b_tow = lambda t: alg[t]["channel"] + alg[t]["token"] * 2          # dgrad of both MLPs + token-mixing wgrad
    emb = sum(alg[t]["embed"] for t in two)
    patches = [                                                       # (object, attribute, name, algorithmic FLOPs per launch)
        (E, "embeds_forward", "embeds_fwd[image+audio]", emb),
        (E, "towers_forward", "towers_fwd[image+audio]", sum(f_tow(t) for t in two) + (emb if getattr(eng, "_embed_fold", False) else 0)),
        (eng.t_fus, "forward", "tower_fwd[fusion]", f_tow("fusion")),
        (E, "heads_ce", "heads_ce", alg["heads"] * 3),
        (eng.t_fus, "backward", "tower_bwd[fusion]", b_tow("fusion")),
        (eng.t_fus, "backward_heads", "tower_bwd[fusion]+heads", b_tow("fusion") + alg["heads"] * 3),
        (E, "towers_backward", "towers_bwd[image+audio]", sum(b_tow(t) for t in two)),
        (E, "towers_wgrad", "towers_wgrad[all+embeds]", sum(alg[t]["channel"] for t in ("image", "audio", "fusion")) + emb),
    ]
    saved = [(o, a, getattr(o, a)) for o, a, _, _ in patches]
    for o, a, name, flops in patches:
        setattr(o, a, timed(name, getattr(o, a), flops))
    try:
        for _ in range(nsteps):
            eng._forward(image, audio, labels, True, True, prologue=True)
            eng._backward(image, audio)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng._update(1.0)                  # the flat Adam launch + the operand re-pack launch (m2m_pack_all); with
                                              # M2M_FUSED_UPDATE=1 one launch (m2m_adam_pack_all)
            e1.record()
            spans.setdefault("adam+pack", {"events": [], "flops": 0})["events"].append((e0, e1))
        torch.cuda.synchronize()
    finally:
        for o, a, orig in saved:
            setattr(o, a, orig)
    out = {}
    for name, sp in spans.items():
        times = [a.elapsed_time(b) * 1e3 for a, b in sp["events"]]      # us
        times = times[len(times) // 5:]                                   # drop the first fifth
        per_launch = float(np.median(times))                             # median: robust to a stray slow launch
        launches_per_step = len(sp["events"]) / nsteps
        out[name] = {"us_per_launch": per_launch, "us_per_step": per_launch * launches_per_step, "flops_per_launch": sp["flops"]}
    return out


if __name__ == "__main__":
    main()
