#!/usr/bin/env python3
"""Phase breakdown (workgroup 0, wave 0) of the split-path kernels with the timers build (make -C m2_mixer_amd/csrc TIMERS=1)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M2M_LIB_PATH", os.path.join(ROOT, "m2_mixer_amd", "libm2mixer_timers.so"))
os.environ["M2M_SPLIT"] = "1"
from m2_mixer_amd import _lib as L          # noqa: E402
from m2_mixer_amd.engine import AVMnistEngine  # noqa: E402
import bench                                   # noqa: E402

lib = L.lib()
NAMES = {
    "scf": ["prologue", "DMA issue+wait+barrier", "compute", "trailing barrier", "epilogue"],
    "scb": ["prologue", "DMA issue+wait+barrier", "compute", "trailing barrier", "epilogue"],
    "smf": ["table+input", "x_in/tokw/LN1", "token mixing", "x_mid/LN2", "operand images"],
    "smb": ["table+carry+slabs", "LN2 bwd", "tokw/LN1 recompute", "token pair loop", "token-grad reduce", "LN1 bwd", "carry/dYd/db2/images"],
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
    batch = bench.make_batch(bench.CFG_B, B, 1234, dev)
    for _ in range(3):
        eng.train_step(*batch)
    torch.cuda.synchronize()
    for k in NAMES:
        read(k)
    n = 10
    for _ in range(n):
        eng.train_step(*batch)
    torch.cuda.synchronize()
    # launches per step of each kind (towers grouped + fusion): chain 4 + 2, mix fwd 5 + 3, mix bwd 5 + 3
    per = {"scf": 6, "scb": 6, "smf": 8, "smb": 8}
    for k, names in NAMES.items():
        raw = read(k, reset=False)
        if raw[19] > 0:
            print(f"{k}: workgroups {raw[19] / (n * per[k]):.0f} per launch, mean workgroup duration {raw[18] / raw[19]:.2f} us "
                  f"(earliest start -> latest end over all launches spans {raw[17] - raw[16]:.0f} us)")
        t = read(k) / (n * per[k])
        t = t[:16]
        print(f"{k}: per launch (WG0 wave0): total {t.sum():.2f} us: " + ", ".join(f"{nm} {v:.2f}" for nm, v in zip(names, t)))


if __name__ == "__main__":
    main()
