#!/usr/bin/env python3
"""Parts of the MM-IMDb / MIMIC step in isolation (HIP events, eager calls after two real steps): TASK=mmimdb|mimic B=..."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_util as G
from wgrad_probe import timeit
from m2_mixer_amd.engine import MMIMDBEngine, MimicEngine
from m2_mixer_amd.runtime import towers_wgrad, embeds_wgrad, pack_all
task, B = os.environ.get("TASK", "mmimdb"), int(os.environ.get("B", "32"))
dev = torch.device("cuda:0")
if task == "mmimdb":
    eng = MMIMDBEngine(dict(G.MMIMDB), B, device=dev, precision="bf16", lr=1e-3, seed=42)
    batch = tuple(t.to(dev) for t in G.mmimdb_batch(B, 1234, G.MMIMDB))
else:
    eng = MimicEngine(dict(G.MIMIC_H), B, device=dev, precision="bf16", lr=1e-2, seed=42)
    batch = tuple(t.to(dev) for t in G.mimic_batch(B, 1234, G.MIMIC_H))
for _ in range(2): eng.train_step(*batch)
torch.cuda.synchronize()
sd = eng.drop_step
if task == "mmimdb":
    tw = [eng.t_fus, eng.t_a, eng.t_b]; em = [eng.e_a, eng.e_b]; inp = list(batch[:2]); dx = [eng.dx0_a, eng.dx0_b]
    print("towers wgrad alone %.1f us" % timeit(lambda: towers_wgrad(tw, B)))
    for t, n in zip(tw, ("fusion", "a", "b")):
        print("  %s alone (group launch) %.1f us, single-tower launch %.1f us" % (n, timeit(lambda: towers_wgrad([t], B)), timeit(lambda: t.wgrad(B, 1, 0, sd))))
    print("embeds wgrad alone %.1f us" % timeit(lambda: embeds_wgrad(em, inp, dx, B)))
    for e, x, d, n in zip(em, inp, dx, "ab"):
        print("  embed %s alone %.1f us" % (n, timeit(lambda: e.wgrad(x, d, B))))
    print("merged %.1f us" % timeit(lambda: towers_wgrad(tw, B, em, inp, dx)))
    print("pack_all %.1f us" % timeit(lambda: pack_all([eng.t_a, eng.t_b, eng.t_fus], em)))
print("adam (whole flat buffer) %.1f us" % timeit(lambda: eng._adam(0, eng.n_params, 1.0, False) if hasattr(eng, "n_params") else None))
