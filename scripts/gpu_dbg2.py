import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_util as G
from m2_mixer_amd.engine import AVMnistEngine
dev = torch.device("cuda:0")
cfg = dict(G.AVMNIST["B"]); B = 64
mode = sys.argv[1]
eng = AVMnistEngine(cfg, B, precision="bf16", lr=1e-3)
# re-home losses into the middle of a guard buffer
guard = torch.zeros(256, device=dev)
eng.losses = guard[128:132]
image, audio, labels = (t.to(dev) for t in G.avmnist_batch(B, 5, cfg))
if mode == "graph":
    replay = eng.capture(image, audio, labels)
    stepf = lambda: replay()
else:
    stepf = lambda: eng.train_step(image, audio, labels)
for i in range(4):
    stepf(); torch.cuda.synchronize()
    gz = guard.cpu()
    nz = [(int(j), float(gz[j])) for j in torch.nonzero(gz).flatten()]
    print(mode, i, nz, flush=True)
