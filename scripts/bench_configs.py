#!/usr/bin/env python3
"""Secondary configurations of SURVEY.md section 8d measured the same way as bench.py (hipGraph replay of the whole
training step, synthetic data, bf16): MIMIC-H (cfg batch 128 and a large-batch point) and MM-IMDb (cfg batch 32 per GPU
and a large-batch point).  One JSON line per configuration.  Not the headline: bench.py is.

    python scripts/bench_configs.py [--steps 50] [--warmup 10] [--precision bf16] [--gpus N] [--only mimic|mmimdb]

--gpus N > 1 (BASELINE configs 3 / 5 scaled: MM-IMDb on 4 GPUs, cfg batch 32 per GPU): one rank per GPU, started through
torch.distributed.run before any GPU call exactly as bench.py does (bench.launch_ranks; or by the caller: WORLD_SIZE set),
gradients exchanged in fp32 by one all-reduce per step (parallel.GradSync), barrier + synchronize on both sides of the timed
region, MAX over ranks, `value` = whole-job samples/s (weak scaling: per-GPU batch fixed).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import gen_util as G  # noqa: E402  (configs + synthetic batches)

MFLOP_PER_SAMPLE = {"mimic": 2.988, "mmimdb": 686.923}      # fwd + bwd, SURVEY.md section 8a


def run(task, B, steps, warmup, precision, dev, world=1, rank=0):
    from m2_mixer_amd import parallel
    from m2_mixer_amd.engine import MimicEngine, MMIMDBEngine
    if task == "mimic":
        cfg = dict(G.MIMIC_H)
        eng = MimicEngine(cfg, B, device=dev, precision=precision, lr=1e-2, seed=42)
        batch = G.mimic_batch(B, parallel.shard_batch_seed(1234, rank), cfg)
    else:
        cfg = dict(G.MMIMDB)
        eng = MMIMDBEngine(cfg, B, device=dev, precision=precision, lr=1e-3, seed=42)
        batch = G.mmimdb_batch(B, parallel.shard_batch_seed(1234, rank), cfg)
    batch = tuple(t.to(dev) for t in batch)
    sync = None
    if world > 1:
        parallel.broadcast_parameters(eng.flat_p)
        eng.pack()
        sync = parallel.GradSync()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    replay = eng.capture(*batch, grad_sync=sync)
    for _ in range(warmup):
        replay()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        replay()
    barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    sps = world * B * steps / dt
    peak = 2500.0 if precision == "bf16" else 157.3           # dense MFMA peak, TFLOP/s (MI355X_MICROARCH.md)
    ach = sps / world * MFLOP_PER_SAMPLE[task] * 1e6 / 1e12      # per GPU
    return {"roofline": {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 5),
                         "traffic": None, "note": "whole step: algorithmic FLOPs (SURVEY.md section 8a) / step time"},
            "metric": f"training samples/sec {task} {precision}", "value": round(sps, 1), "unit": "samples/s",
            "ms_per_step": round(dt / steps * 1e3, 4), "batch": B, "steps": steps, "warmup": warmup, "dtype": precision,
            "n_gpus": world, "scaling": "weak", "grad_allreduce": "fp32" if world > 1 else None,
            "data": "synthetic", "n_params": eng.n_params,
            "achieved_tflops": round(ach, 3), "final_loss": round(float(eng.losses[3]), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--only", default=None, choices=[None, "mimic", "mmimdb"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--cfg-batch-only", action="store_true", help="only the cfg batch of each task (the multi-GPU runs)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import bench                                              # nothing has touched the GPU yet in this process
        # (launch_ranks relays exactly one JSON line; here every configuration prints one: run them one per launch)
        rc = 0
        for task in ("mimic", "mmimdb"):
            if args.only in (None, task):
                rc |= bench.launch_ranks(args.gpus, [a for a in sys.argv[1:] if a not in ("mimic", "mmimdb", "--only")] +
                                         ["--only", task, "--cfg-batch-only"], script=os.path.abspath(__file__))
        sys.exit(rc)
    from m2_mixer_amd import parallel
    rank, local_rank, world = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device(f"cuda:{int(os.environ.get('M2M_FORCE_DEVICE', local_rank))}")
    torch.cuda.set_device(dev)
    for task, B in (("mimic", 128), ("mimic", 8192), ("mmimdb", 32), ("mmimdb", 256)):
        if args.only and task != args.only:
            continue
        if args.cfg_batch_only and B not in (128, 32):
            continue
        out = run(task, B, args.steps, args.warmup, args.precision, dev, world, rank)
        if rank == 0:
            print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
