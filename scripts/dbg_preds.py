"""bf16 AV-MNIST step vs oracle: logit errors, argmax flips and the oracle's top-2 margin at the flipped samples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import gen_util as G
from oracle import m2mixer_oracle as O
from m2_mixer_amd.engine import AVMnistEngine
dev = torch.device("cuda:0")
for size, B in (("S", 8), ("B", 40), ("B", 13)):
    cfg = dict(G.AVMNIST[size], dropout=0.0)
    eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-2, init=False)
    shapes = G.avmnist_shapes(cfg)
    params = dict(G.make_params(shapes, 7))
    eng.load_state_dict(params)
    image, audio, labels = G.avmnist_batch(B, 8, cfg)
    eng.forward_backward(image.to(dev), audio.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    ref = O.avmnist_train_step(image, audio, labels, params, cfg, {}, lr=1e-2)
    for i, k in ((2, "logits"), (0, "image_logits"), (1, "audio_logits")):
        lg = eng.logits[i].cpu().float(); rf = ref[k]
        err = (lg - rf).abs().max().item()
        flips = (lg.argmax(1) != rf.argmax(1)).nonzero().flatten().tolist()
        top2 = rf.topk(2, dim=1).values
        print(size, B, k, "max abs err %.4g" % err, "flips", flips, "margins", [(top2[f, 0] - top2[f, 1]).item() for f in flips])
    worst = max(((eng.grads[k].cpu() - g).norm() / (g.norm() + 1e-12)).item() for k, g in ref["grads"].items() if not k.endswith("token_mix.2.net.3.bias"))
    print(size, B, "worst grad relerr %.4g" % worst)
