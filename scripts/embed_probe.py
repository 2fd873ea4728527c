#!/usr/bin/env python3
"""Time the patch-embedding forward launches in isolation (HIP events): audio alone, image alone, and the grouped launch
with the audio k-split.  M2M_LIB_PATH selects the library build (ablation builds: make EXP=...)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench
from wgrad_probe import timeit
from m2_mixer_amd.engine import AVMnistEngine
from m2_mixer_amd.runtime import embeds_forward
dev = torch.device("cuda:0"); B = 512
eng = AVMnistEngine(bench.CFG_B, B, device=dev, precision="bf16", lr=1e-2)
batch = bench.make_batch(bench.CFG_B, B, 1234, dev)
for _ in range(2): eng.train_step(*batch)
torch.cuda.synchronize()
print(os.path.basename(os.environ.get("M2M_LIB_PATH", "default")),
      "audio %.1f us" % timeit(lambda: eng.e_b.forward(batch[1], B, eng.x0_b)),
      "image %.1f us" % timeit(lambda: eng.e_a.forward(batch[0], B, eng.x0_a)),
      "both %.1f us" % timeit(lambda: embeds_forward([eng.e_a, eng.e_b], list(batch[:2]), [eng._x0_a, eng._x0_b], B, list(eng.x0_splits))))
