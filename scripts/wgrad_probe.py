#!/usr/bin/env python3
"""Time the weight-gradient launches in isolation (HIP events): one tower alone (96 workgroups: one per CU, no
co-residency) and the merged three-tower launch.  M2M_LIB_PATH selects the library build (ablation builds: make EXP=...)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from m2_mixer_amd.engine import AVMnistEngine  # noqa: E402
from m2_mixer_amd.runtime import towers_wgrad  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    eng = AVMnistEngine(bench.CFG_B, B, device=dev, precision="bf16", lr=1e-2)
    batch = bench.make_batch(bench.CFG_B, B, 1234, dev)
    for _ in range(2):
        eng.train_step(*batch)
    torch.cuda.synchronize()
    sd = eng.drop_step
    print(f"B {B} lib {os.path.basename(os.environ.get('M2M_LIB_PATH', 'libm2mixer.so'))}: "
          f"image alone {timeit(lambda: eng.t_img.wgrad(B, 1, 0, sd)):.1f} us, "
          f"fusion alone {timeit(lambda: eng.t_fus.wgrad(B, 1, 0, sd)):.1f} us, "
          f"image+audio {timeit(lambda: towers_wgrad([eng.t_a, eng.t_b], B)):.1f} us, "
          f"merged {timeit(lambda: towers_wgrad([eng.t_fus, eng.t_a, eng.t_b], B)):.1f} us, "
          f"merged + embeds {timeit(lambda: towers_wgrad([eng.t_fus, eng.t_a, eng.t_b], B, [eng.e_a, eng.e_b], list(batch[:2]), [eng.dx0_a, eng.dx0_b])):.1f} us")


if __name__ == "__main__":
    main()
