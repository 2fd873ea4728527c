import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_util as G
import m2_mixer_amd as M
from m2_mixer_amd import modules as MM, _lib as L
from m2_mixer_amd.runtime import TowerRuntime, BLOCK_FIELDS, BLOCK_KEYS
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
case = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (4, 32, 16, 256)
N, D, T, Cc = case
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p, x, dy = G.block_case_tensors(case, B, seed=1)
M.set_precision(prec)
rt = TowerRuntime(D, N, T, Cc, 1, False, 0.0, M.config.prec_id(), 0)
P = {f: p[BLOCK_KEYS[f]].to(dev).contiguous() for f in BLOCK_FIELDS}
rt.bind_params([P], None)
print("bound", flush=True)
rt.pack(); torch.cuda.synchronize(); print("packed", flush=True)
xg = x.to(dev); out = torch.empty_like(xg)
rt.forward(xg, N * D, B, out, N * D, None, False, 0, 0); torch.cuda.synchronize(); print("fwd eval ok", float(out.abs().max()), flush=True)
rt.forward(xg, N * D, B, out, N * D, None, True, 0, 0); torch.cuda.synchronize(); print("fwd train ok", float(out.abs().max()), flush=True)
flat = torch.zeros(rt.grad_numel(), device=dev); rt.bind_grads(flat)
dx = torch.empty_like(xg)
rt.backward(B, dy.to(dev), N * D, None, dx, N * D, 0, 0); torch.cuda.synchronize(); print("bwd ok", float(dx.abs().max()), flush=True)
rt.wgrad(B, 0, 0); torch.cuda.synchronize(); print("wgrad ok", float(flat.abs().max()), flush=True)
