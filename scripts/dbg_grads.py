"""Dump the gradients of one bf16 M2-Mixer-B training step (dropout 0.5, B = 40) for the library M2M_LIB_PATH names; with two
dumps given, compare them tensor by tensor:  python scripts/dbg_grads.py dump out.pt | python scripts/dbg_grads.py cmp a.pt b.pt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
if sys.argv[1] == "dump":
    import gen_util as G
    from m2_mixer_amd.engine import AVMnistEngine
    dev = torch.device("cuda:0")
    cfg = dict(G.AVMNIST["B"], dropout=float(os.environ.get("P", "0.5")))
    B = int(os.environ.get("B", "40"))
    eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, init=False)
    eng.load_state_dict(dict(G.make_params(G.avmnist_shapes(cfg), 17)))
    image, audio, labels = G.avmnist_batch(B, 18, cfg)
    eng.forward_backward(image.to(dev), audio.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu().clone() for k, v in eng.grads.items()}, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k] - b[k]).abs().max().item(); s = b[k].abs().max().item()
        flag = "  <<<<" if d > 0.05 * max(s, 1e-12) else ""
        print(f"{k:60s} maxdiff {d:.3e}  ref max {s:.3e}{flag}")
