"""Eager training steps of a secondary configuration for rocprofv3 --kernel-trace: TASK=mimic|mmimdb B=..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_util as G  # noqa: E402
from m2_mixer_amd.engine import MimicEngine, MMIMDBEngine  # noqa: E402

task, B = os.environ.get("TASK", "mmimdb"), int(os.environ.get("B", "32"))
dev = torch.device("cuda:0")
if task == "mimic":
    cfg = dict(G.MIMIC_H)
    eng = MimicEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, seed=42)
    batch = G.mimic_batch(B, 1234, cfg)
else:
    cfg = dict(G.MMIMDB)
    eng = MMIMDBEngine(cfg, B, device=dev, precision="bf16", lr=1e-3, seed=42)
    batch = G.mmimdb_batch(B, 1234, cfg)
batch = tuple(t.to(dev) for t in batch)
replay = eng.capture(*batch)
for _ in range(int(os.environ.get("STEPS", "30"))):
    replay()
torch.cuda.synchronize()
print("done", float(eng.losses[3]))
