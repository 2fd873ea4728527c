#!/usr/bin/env python3
"""Parts of the merged weight-gradient launch in isolation (HIP events): the three towers alone, both patch embeddings
alone, everything merged.  M2M_LIB_PATH selects the library build."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench
from wgrad_probe import timeit
from m2_mixer_amd.engine import AVMnistEngine
from m2_mixer_amd.runtime import towers_wgrad, embeds_wgrad
dev = torch.device("cuda:0"); B = 512
eng = AVMnistEngine(bench.CFG_B, B, device=dev, precision="bf16", lr=1e-2)
batch = bench.make_batch(bench.CFG_B, B, 1234, dev)
for _ in range(2): eng.train_step(*batch)
torch.cuda.synchronize()
tw = [eng.t_fus, eng.t_a, eng.t_b]
em = [eng.e_a, eng.e_b]; inp = list(batch[:2]); dx = [eng.dx0_a, eng.dx0_b] if hasattr(eng, "dx0_a") else None
print("towers alone %.1f us" % timeit(lambda: towers_wgrad(tw, B)))
for t, n in ((eng.t_fus, "fusion"), (eng.t_a, "image"), (eng.t_b, "audio")):
    print(" ", n, "alone %.1f us" % timeit(lambda: towers_wgrad([t], B)))
if dx is not None:
    print("embeds alone %.1f us" % timeit(lambda: embeds_wgrad(em, inp, dx, B)))
    print("merged (row-group embeds / separate launch) %.1f us" % timeit(lambda: towers_wgrad(tw, B, em, inp, dx)))
    if getattr(eng, "_embed_towers", None):
        print("merged (fast embeds) %.1f us" % timeit(lambda: towers_wgrad(tw, B, em, inp, dx, embed_towers=eng._embed_towers)))
else:
    print([k for k in vars(eng) if "dx" in k or "d_x" in k])
