"""Whole-step hipGraph capture for the MODULE path (the reference's own API: `shared_step` -> `loss.backward()` ->
`optimizer.step()`, models/avmnist.py:236-312, :413-415).

The engines (engine.py) capture their own fused steps.  A user who only swapped `modules` / `models` for `m2_mixer_amd.modules` /
`m2_mixer_amd.models` runs ~150 eager launches per step and is host-bound (3-4 ms per step for M2-Mixer-B at batch 512); the same
step replayed as ONE graph takes ~2 ms (bench.py: `module_path_graphed`).  What capture needs and this class arranges:

* dropout step counters on the device (config.set_device_dropout_step): a host integer would be baked into the graph;
* torch.optim.Adam / AdamW with `capturable=True` (its step count lives on the device) and, to change the learning rate without
  re-capturing, `lr` as a device tensor: use `GraphedStep.set_lr`;
* static input buffers: `__call__(batch)` copies the batch into them (same shapes as the example batch) and replays;
* the warm-up steps capture needs are UNDONE (parameters, optimizer state): constructing a GraphedStep does not train.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import config


class GraphedStep:
    def __init__(self, net: torch.nn.Module, optimizer: torch.optim.Optimizer, example_batch: Dict[str, torch.Tensor],
                 step_fn: Optional[Callable] = None, warmup: int = 2, fused: bool = True):
        """step_fn(net, batch) -> dict with a "loss" entry (default: net.shared_step(batch, mode="train")).  The optimizer must not
        have taken a step yet unless it was built with capturable=True."""
        if not isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)):
            raise TypeError("GraphedStep: torch.optim.Adam / AdamW (capturable) only")
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedStep: the model must live on the GPU")
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                if any(len(optimizer.state.get(p, {})) for p in g["params"]):
                    raise RuntimeError("GraphedStep: this optimizer has already stepped without capturable=True; build it with capturable=True")
                g["capturable"] = True
            # the fused implementation (one multi-tensor kernel instead of ~12 foreach passes over 33 MB): 2.06 -> 1.3 ms per
            # replayed step of M2-Mixer-B.  (It does not advance the parameters' version counters; the modules re-pack their
            # operand copies unconditionally in training mode: modules/mixer.py, _train_repack.)
            if fused and all(p.is_cuda and torch.is_floating_point(p) for p in g["params"]):
                g["fused"], g["foreach"] = True, False
            if not torch.is_tensor(g["lr"]):
                g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)      # replays read the CURRENT value
        config.set_device_dropout_step(True)
        self.net, self.optimizer = net, optimizer
        self._step_fn = step_fn or (lambda n, b: n.shared_step(b, mode="train"))
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}

        def one_step():
            optimizer.zero_grad(set_to_none=True)
            out = self._step_fn(net, self.static)
            out["loss"].backward()
            optimizer.step()
            return out

        # snapshot (the warm-up below takes real optimisation steps; everything is restored IN PLACE: the graph holds the addresses)
        params = [p for g in optimizer.param_groups for p in g["params"]]
        snap_p = [p.detach().clone() for p in params]
        had_state = any(len(optimizer.state.get(p, {})) for p in params)
        snap_s = [{k: v.detach().clone() for k, v in optimizer.state.get(p, {}).items() if torch.is_tensor(v)} for p in params]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                one_step()                                      # (the first one creates the optimizer state: device step counters)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.out = one_step()
        with torch.no_grad():
            for p, sp in zip(params, snap_p):
                p.copy_(sp)
            for p, ss in zip(params, snap_s):
                for k, v in optimizer.state[p].items():
                    if torch.is_tensor(v):
                        # a fresh optimizer's state is all zeros (moments, step count); a resumed one gets its values back
                        v.copy_(ss[k]) if (had_state and k in ss) else v.zero_()
        torch.cuda.synchronize(dev)

    def set_lr(self, lr: float) -> None:
        """Change the learning rate of every parameter group without re-capturing (ReduceLROnPlateau: call this from its hook)."""
        for g in self.optimizer.param_groups:
            g["lr"].fill_(float(lr))

    def __call__(self, batch: Dict[str, torch.Tensor]) -> dict:
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.out
