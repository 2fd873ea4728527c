#!/usr/bin/env python3
"""Can the IMPORT-SWAP path (m2_mixer_amd.models / m2_mixer_amd.modules under torch autograd + torch.optim.Adam) be replayed as ONE
hipGraph?  Whole-network capture (torch.cuda.CUDAGraph): shared_step -> loss.backward() -> Adam(capturable=True).step(); checks
the replayed step against the eager one and times both."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
import m2_mixer_amd as M                       # noqa: E402
from m2_mixer_amd import models as MD          # noqa: E402


def build(cfg, B, dev, precision):
    M.set_precision(precision)
    mods = {"image": dict(cfg["image"], block_type="MLPMixer"), "audio": dict(cfg["audio"], block_type="MLPMixer"),
            "multimodal": dict(cfg["multimodal"], block_type="FusionMixer", fusion_function="ConcatFusion"),
            "classification": dict(classifier="StandardClassifier", num_classes=cfg["num_classes"],
                                   input_shape=[B, bench.n_patch(cfg["image"]) + bench.n_patch(cfg["audio"]), cfg["multimodal"]["hidden_dim"]])}
    torch.manual_seed(42)
    net = MD.AVMnistMixerMultiLoss({"dropout": cfg["dropout"], "modalities": mods}, {"lr": 1e-2, "betas": (0.9, 0.999), "scheduler_patience": 2}).to(dev)
    net.train()
    opt = net.configure_optimizers()["optimizer"]
    return net, opt


def main():
    dev = torch.device("cuda:0")
    cfg, B, precision = bench.CFG_B, 512, "bf16"
    image, audio, labels = bench.make_batch(cfg, B, 1234, dev)
    batch = {"image": image, "audio": audio, "label": labels}

    def timed(fn, n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    net, opt = build(cfg, B, dev, precision)
    for g in opt.param_groups:
        g["capturable"] = True
        if len(sys.argv) > 1 and sys.argv[1] == "fused":
            g["fused"], g["foreach"] = True, False

    def step():
        opt.zero_grad(set_to_none=True)
        out = net.shared_step(batch, mode="train")
        out["loss"].backward()
        opt.step()
        return out["loss"]

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print(f"eager: {timed(step, 50):.3f} ms per step", flush=True)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = step()
    torch.cuda.synchronize()
    l0 = float(loss)
    graph.replay(); torch.cuda.synchronize()
    print(f"captured; loss at capture {l0:.4f}, after a replay {float(loss):.4f}", flush=True)
    print(f"graph replay: {timed(graph.replay, 200):.3f} ms per step, loss now {float(loss):.4f}", flush=True)


if __name__ == "__main__":
    main()
