#!/usr/bin/env python3
"""Phase breakdown of the tower kernels with the diagnostic timers build (make -C m2_mixer_amd/csrc TIMERS=1).
Run with M2M_LIB_PATH=m2_mixer_amd/libm2mixer_timers.so on the GPU box."""
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
import bench                                   # noqa: E402

lib = L.lib()
NAMES = {
    "fwd": ["block input", "token mix", "save/LN2/pack", "column loop", "slabs", "sum+residual(+next LN1)", "final LN/out"],
    "bwd": ["params+upstream+LNf bwd", "X1 dYd/A/images", "C3 column loop", "C4 slabs(+wait)", "R1 sum/LN2bwd/LN1/operands", "R2 token MLP bwd + LN2 colsums", "R3 token atomics + LN1 bwd"],
    "wgrad": ["wait loads + stage write", "issue refill loads", "barrier", "LDS reads + MFMA", "write-out"],
}


def read(kind, reset=True):
    buf = (C.c_ulonglong * 32)()
    fn = getattr(lib, f"m2m_debug_timers_{kind}")
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, int(reset)) == 0
    return np.array(list(buf), dtype=np.float64) * 0.01   # 100 MHz ticks -> us


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    eng = AVMnistEngine(bench.CFG_B, B, device=dev, precision="bf16", lr=1e-2)
    image, audio, labels = bench.make_batch(bench.CFG_B, B, 1234, dev)
    n = 5
    for _ in range(2):
        eng.train_step(image, audio, labels)
    torch.cuda.synchronize()
    for kind in ("fwd", "bwd", "wgrad"):
        read(kind)
    for name, rt, x0, N in (("image", eng.t_img, eng.x0_img, eng.Ni), ("fusion", eng.t_fus, eng.fused, eng.Nf)):
        D = eng.D
        out = torch.empty(B, N, D, device=dev)
        for _ in range(n):
            rt.forward(x0, N * D, B, out, N * D, None, True, 1, 0, eng.drop_step)
        torch.cuda.synchronize()
        t = read("fwd") / n
        print(f"tower_fwd[{name}] per launch (WG0): total {t.sum():.1f} us: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(NAMES['fwd'], t)))
        dout = torch.randn(B, N, D, device=dev)
        dx = torch.empty(B, N, D, device=dev)
        for _ in range(n):
            rt.backward(B, dout, N * D, None, dx, N * D, 1, 0, eng.drop_step)
        torch.cuda.synchronize()
        t = read("bwd") / n
        print(f"tower_bwd[{name}] per launch (WG0): total {t.sum():.1f} us: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(NAMES['bwd'], t)))
        for _ in range(n):
            rt.wgrad(B, 1, 0, eng.drop_step)
        torch.cuda.synchronize()
        t = read("wgrad") / n
        print(f"tower_wgrad[{name}] per launch (WG0): total {t.sum():.1f} us: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(NAMES['wgrad'], t)))
    from m2_mixer_amd.runtime import towers_wgrad
    for _ in range(n):
        towers_wgrad([eng.t_fus, eng.t_a, eng.t_b], B)
    torch.cuda.synchronize()
    t = read("wgrad") / n
    print(f"towers_wgrad[all three] per launch (WG0): total {t.sum():.1f} us: " + ", ".join(f"{k} {v:.1f}" for k, v in zip(NAMES['wgrad'], t)))


if __name__ == "__main__":
    main()
