#!/usr/bin/env python3
"""Phase timers of the recompute-form weight-gradient launch (diagnostic build: make -C m2_mixer_amd/csrc TIMERS=1)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M2M_LIB_PATH", os.path.join(ROOT, "m2_mixer_amd", "libm2mixer_timers.so"))
from m2_mixer_amd import _lib as L
from m2_mixer_amd.engine import AVMnistEngine
from m2_mixer_amd.runtime import towers_wgrad
import bench
lib = L.lib()
def read(reset=True):
    buf = (C.c_ulonglong * 32)()
    fn = lib.m2m_debug_timers_wgrad; fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, int(reset)) == 0
    return np.array(list(buf), dtype=np.float64)
dev = torch.device("cuda:0"); B = 512
eng = AVMnistEngine(bench.CFG_B, B, device=dev, precision="bf16", lr=1e-2)
batch = bench.make_batch(bench.CFG_B, B, 1234, dev)
for _ in range(3): eng.train_step(*batch)
torch.cuda.synchronize()
names = ["prologue", "dma wait", "barrier", "dma issue", "compute", "write-out"]
def run(label, fn, n=5):
    read()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    t = read()
    ph = t[:6] * 0.01 / n
    cnt = max(t[19], 1)
    print(f"{label}: launch {e0.elapsed_time(e1) * 1e3 / n:.1f} us | WG0 phases (us): " + ", ".join(f"{k} {v:.1f}" for k, v in zip(names, ph)) +
          f" | WGs per launch {cnt / n:.0f}, mean WG duration {t[18] / cnt * 0.01:.1f} us, span first start -> last end {(t[17] - t[16]) * 0.01:.1f} us (last of {n}),"
          f" latest start - earliest start {(t[20] - t[16]) * 0.01:.1f} us")
kw = dict(seed=eng.seed, step=0, step_dev=eng.drop_step)
tw = [eng.t_fus, eng.t_a, eng.t_b]
run("towers only", lambda: towers_wgrad(tw, B, **kw))
run("towers + embeds", lambda: towers_wgrad(tw, B, [eng.e_a, eng.e_b], list(batch[:2]), [eng.dx0_a, eng.dx0_b], **kw))
run("image tower only", lambda: towers_wgrad([eng.t_a], B, **kw))
run("fusion tower only", lambda: towers_wgrad([eng.t_fus], B, **kw))
