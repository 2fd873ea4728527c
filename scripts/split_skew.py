#!/usr/bin/env python3
"""Start / end skew of the workgroups of ONE chain-forward launch (timers build)."""
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


def read(kind, reset=True):
    buf = (C.c_ulonglong * 32)()
    fn = getattr(lib, f"m2m_debug_timers_{kind}")
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, int(reset)) == 0
    return [int(v) for v in buf]


dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = dict(bench.CFG_B, multimodal=dict(bench.CFG_B["multimodal"], num_mixers=1))
eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2)
batch = bench.make_batch(bench.CFG_B, B, 1234, dev)
for _ in range(3):
    eng.train_step(*batch)
torch.cuda.synchronize()
# one fusion-tower forward = 3 mix + 2 chain launches; read after each whole call, chain launches only touch "scf"
for rep in range(4):
    read("scf")
    eng.t_fus.forward(eng.fused, eng.Nf * eng.D, B, eng.fus_out, eng.Nf * eng.D, eng.pool_fus, True, eng.seed, 0, eng.drop_step)
    torch.cuda.synchronize()
    r = read("scf")
    n = r[19]
    print(f"fusion forward (1 chain launch, {n} workgroups): mean workgroup {r[18] / n * 0.01:.2f} us; launch spans "
          f"{(r[17] - r[16]) * 0.01:.2f} us; latest start - earliest start {(r[20] - r[16]) * 0.01:.2f} us; "
          f"latest end - earliest end {(r[17] - r[21]) * 0.01:.2f} us")
