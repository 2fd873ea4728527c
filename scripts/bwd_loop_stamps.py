#!/usr/bin/env python3
"""Stage breakdown of ONE wave's hidden-column step in the backward chain kernel (wave 0 of workgroup 0), in shader cycles
(s_memtime stamps of the timers build: make -C m2_mixer_amd/csrc TIMERS=1; TIMER_CMARK 8..14 in tower_bwd.hip).  The stamps
serialize the wave at every stage boundary (s_memtime returns through lgkmcnt): read the split, not the total."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M2M_LIB_PATH", os.path.join(ROOT, "m2_mixer_amd", "libm2mixer_timers.so"))
from m2_mixer_amd import _lib as L          # noqa: E402
from m2_mixer_amd.engine import AVMnistEngine  # noqa: E402
import m2_mixer_amd.engine as E             # noqa: E402
import bench                                   # noqa: E402

lib = L.lib()
STAGES = ["head (ticket, bias, acc init)", "weights wait + W1 park + A/dYd reads + products 1-2 issue", "prefetch issue",
          "epilogue (products done, keep-words, table, chain frags)", "transposes + packs + operand stores", "third product"]


def read(reset=True):
    buf = (C.c_ulonglong * 32)()
    fn = lib.m2m_debug_timers_bwd
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, int(reset)) == 0
    return np.array(list(buf), dtype=np.float64)


def report(name, t, n_launch, nblocks, steps_per_wave):
    steps = n_launch * nblocks * steps_per_wave
    cyc = t[8:14] / steps
    loop_ticks = t[2] / n_launch / nblocks          # 100 MHz ticks per block loop (slot 2: the phase timer of the loop)
    loop_cyc = t[8:15].sum() / n_launch / nblocks
    mhz = loop_cyc / max(loop_ticks, 1e-9) * 100.0
    print(f"{name}: per step {cyc.sum():.0f} cycles | " + " | ".join(f"{s}: {c:.0f}" for s, c in zip(STAGES, cyc)))
    print(f"{name}: loop per block {loop_ticks * 0.01:.2f} us = {loop_cyc:.0f} cycles -> shader clock {mhz:.0f} MHz; phases (us per launch): "
          + ", ".join(f"{v * 0.01 / n_launch:.1f}" for v in t[0:7]))


def main():
    dev = torch.device("cuda:0")
    B = 512
    eng = AVMnistEngine(bench.CFG_B, B, device=dev, precision="bf16", lr=1e-2)
    image, audio, labels = bench.make_batch(bench.CFG_B, B, 1234, dev)
    for _ in range(3):
        eng.train_step(image, audio, labels)
    torch.cuda.synchronize()
    read()
    n = 10
    D = eng.D
    for name, rt, N in (("image alone (128 WGs)", eng.t_img, eng.Ni), ("fusion (256 WGs)", eng.t_fus, eng.Nf)):
        dout = torch.randn(B, N, D, device=dev)
        dx = torch.empty(B, N, D, device=dev)
        for _ in range(n):
            rt.backward(B, dout, N * D, None, dx, N * D, 1, 0, eng.drop_step)
        torch.cuda.synchronize()
        report(name, read(), n, rt.nblocks, -(-(rt.desc.Cp // 32) // 8))
    # the two-tower launch of the training step (workgroup 0 = image tower, tile 0)
    fs = eng.Nf * D
    d_b_part = eng.d_fused.view(-1)[eng.Na * D:]
    for _ in range(n):
        E.towers_backward([eng.t_a, eng.t_b],
                          [(eng.d_fused, fs, eng.dpool_a, eng.dx0_a, eng.Na * D), (d_b_part, fs, eng.dpool_b, eng.dx0_b, eng.Nb * D)],
                          B, eng.seed, 0, eng.drop_step)
    torch.cuda.synchronize()
    report("image+audio launch (256 WGs)", read(), n, eng.t_a.nblocks, -(-(eng.t_a.desc.Cp // 32) // 8))


if __name__ == "__main__":
    main()
