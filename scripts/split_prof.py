"""Run the training step eagerly a few dozen times (for rocprofv3 --kernel-trace --stats): M2M_SPLIT from the environment."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_util as G  # noqa: E402
from m2_mixer_amd.engine import AVMnistEngine  # noqa: E402

B = int(os.environ.get("B", "512"))
dev = torch.device("cuda:0")
cfg = dict(G.AVMNIST["B"])
eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-3, seed=3)
batch = tuple(t.to(dev) for t in G.avmnist_batch(B, 5, cfg))
for _ in range(int(os.environ.get("STEPS", "40"))):
    eng.train_step(*batch)
torch.cuda.synchronize()
print("done", float(eng.losses[3]))
