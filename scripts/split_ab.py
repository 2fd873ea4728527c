"""A/B of the split path against the fused one-launch towers on the same engine (M2M_SPLIT=0 / 1): agreement of logits /
losses / gradients, and HIP-event timings of the forward and backward tower launches.  Run on the GPU box."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_util as G  # noqa: E402
from m2_mixer_amd.engine import AVMnistEngine  # noqa: E402


def run(B, p_drop, what):
    dev = torch.device("cuda:0")
    cfg = dict(G.AVMNIST["B"], dropout=p_drop)
    eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-3, seed=3)
    batch = tuple(t.to(dev) for t in G.avmnist_batch(B, 5, cfg))
    res = {}
    for mode in ("0", "1"):
        os.environ["M2M_SPLIT"] = mode
        eng.drop_step.zero_(); eng.adam_state[0] = 0; eng.flat_g.zero_()
        if what == "eval":
            out = eng.evaluate(*batch)
            torch.cuda.synchronize()
            res[mode] = {"logits": eng.logits.clone(), "losses": eng.losses.clone()}
        else:
            eng.forward_backward(*batch)
            torch.cuda.synchronize()
            res[mode] = {"logits": eng.logits.clone(), "losses": eng.losses.clone(), "grads": eng.flat_g.clone(),
                         "x_mid": eng.t_a._keep["saved3"]["x_mid"].clone(), "x_final": eng.t_fus._keep["x_final"].clone()}
            eng.flat_g.zero_()
    a, b = res["0"], res["1"]
    msg = [f"B={B} p={p_drop} {what}:"]
    for k in a:
        d = float((a[k].float() - b[k].float()).abs().max())
        s = float(a[k].float().abs().max())
        msg.append(f"{k} maxdiff {d:.3e} (scale {s:.3e})")
    print("  ".join(msg), flush=True)
    if what != "eval":
        # per-tensor relative gradient differences
        worst = []
        for k, gv in eng.grads.items():
            o = (gv.data_ptr() - eng.flat_g.data_ptr()) // 4
            ga, gb = a["grads"][o:o + gv.numel()], b["grads"][o:o + gv.numel()]
            worst.append((float((ga - gb).abs().max()) / (float(ga.abs().max()) + 1e-12), k))
        worst.sort(reverse=True)
        print("   worst grads:", [(f"{w:.2e}", k) for w, k in worst[:4]], flush=True)


def timing(B):
    dev = torch.device("cuda:0")
    cfg = dict(G.AVMNIST["B"])
    eng = AVMnistEngine(cfg, B, device=dev, precision="bf16", lr=1e-3, seed=3)
    batch = tuple(t.to(dev) for t in G.avmnist_batch(B, 5, cfg))
    for mode in ("0", "1"):
        os.environ["M2M_SPLIT"] = mode
        for _ in range(5):
            eng._forward(*batch, training=True, with_grad=True, prologue=True)
            eng._backward(*batch[:-1])
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tf = tb = 0.0
        n = 20
        for _ in range(n):
            ev[0].record()
            eng._forward(*batch, training=True, with_grad=True, prologue=True)
            ev[1].record()
            eng._backward(*batch[:-1])
            ev[2].record()
            torch.cuda.synchronize()
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
        print(f"B={B} M2M_SPLIT={mode}: forward {tf / n * 1e3:.1f} us  backward+wgrad {tb / n * 1e3:.1f} us (eager, HIP events)", flush=True)
        eng.flat_g.zero_()
    # graph-replayed whole step
    for mode in ("0", "1"):
        os.environ["M2M_SPLIT"] = mode
        replay = eng.capture(*batch, steps=10)
        for _ in range(10):
            replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        print(f"B={B} M2M_SPLIT={mode}: {dt * 1e6:.1f} us / step (graph, 10 steps per graph) = {B / dt:.0f} samples/s", flush=True)


if __name__ == "__main__":
    for B, p, what in ((512, 0.0, "eval"), (200, 0.0, "eval"), (512, 0.5, "train"), (200, 0.1, "train"), (512, 0.0, "train")):
        run(B, p, what)
    timing(512)
