#!/usr/bin/env python3
"""Secondary configurations of SURVEY.md section 8d measured the same way as bench.py (hipGraph replay of the whole
training step, synthetic data, bf16): MIMIC-H (cfg batch 128 and a large-batch point) and MM-IMDb (cfg batch 32 per GPU
and a large-batch point).  One JSON line per configuration.  Not the headline: bench.py is.

    python scripts/bench_configs.py [--steps 50] [--warmup 10] [--precision bf16]
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


def run(task, B, steps, warmup, precision, dev):
    from m2_mixer_amd.engine import MimicEngine, MMIMDBEngine
    if task == "mimic":
        cfg = dict(G.MIMIC_H)
        eng = MimicEngine(cfg, B, device=dev, precision=precision, lr=1e-2, seed=42)
        batch = G.mimic_batch(B, 1234, cfg)
    else:
        cfg = dict(G.MMIMDB)
        eng = MMIMDBEngine(cfg, B, device=dev, precision=precision, lr=1e-3, seed=42)
        batch = G.mmimdb_batch(B, 1234, cfg)
    batch = tuple(t.to(dev) for t in batch)
    replay = eng.capture(*batch)
    for _ in range(warmup):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sps = B * steps / dt
    peak = 2500.0 if precision == "bf16" else 157.3           # dense MFMA peak, TFLOP/s (MI355X_MICROARCH.md)
    ach = sps * MFLOP_PER_SAMPLE[task] * 1e6 / 1e12
    return {"roofline": {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 5),
                         "traffic": None, "note": "whole step: algorithmic FLOPs (SURVEY.md section 8a) / step time"},
            "metric": f"training samples/sec {task} {precision}", "value": round(sps, 1), "unit": "samples/s",
            "ms_per_step": round(dt / steps * 1e3, 4), "batch": B, "steps": steps, "warmup": warmup, "dtype": precision,
            "data": "synthetic", "n_params": eng.n_params,
            "achieved_tflops": round(sps * MFLOP_PER_SAMPLE[task] * 1e6 / 1e12, 3), "final_loss": round(float(eng.losses[3]), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--only", default=None, choices=[None, "mimic", "mmimdb"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for task, B in (("mimic", 128), ("mimic", 8192), ("mmimdb", 32), ("mmimdb", 256)):
        if args.only and task != args.only:
            continue
        print(json.dumps(run(task, B, args.steps, args.warmup, args.precision, dev)), flush=True)


if __name__ == "__main__":
    main()
