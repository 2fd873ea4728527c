"""Plain ReLU MLP, the MIMIC `static` tower (reference: modules/mlp.py:4-27): 5 -> 64 -> 64 -> 64,
~8.5 kMAC per sample.  module_list indices follow the reference (Linear at 3*i, output Linear at 3*num_blocks) so
checkpoints load; the ReLU / Dropout entries are parameter-free placeholders.  forward() runs csrc/mlp.hip
(m2m_mlp_forward / m2m_mlp_backward) through a torch.autograd.Function; CPU tensors raise."""
from __future__ import annotations

import itertools
from typing import Optional

import torch
from torch import nn

from .. import _lib as L
from .. import config
from ..runtime import MlpRuntime

_site_counter = itertools.count(0)


class _MlpFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, owner, training, *params):
        rt: MlpRuntime = owner._rt
        B = x.shape[0]
        out = torch.empty(B, rt.dims[-1], device=x.device, dtype=torch.float32)
        dropping = training and owner.dropout_p > 0
        rt.desc.p_drop = float(owner.dropout_p) if dropping else 0.0
        step, step_dev = owner._step_args(x.device) if dropping else (0, None)
        ctx.acts = rt.fresh_acts(B, x.device)        # this forward's own saved activations: the module path is re-entrant
        rt.forward(x, B, out, rt.dims[-1], None, True, config.dropout_seed(), step, step_dev)     # training=True: keep the activations
        ctx.owner, ctx.B = owner, B
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        owner = ctx.owner
        rt: MlpRuntime = owner._rt
        (x,) = ctx.saved_tensors
        grads = [torch.zeros_like(p) for p in owner._linear_params()]
        rt.bind(owner._param_pairs(), [(grads[2 * i], grads[2 * i + 1]) for i in range(rt.nlayers)], ctx.B)
        rt.use_acts(ctx.acts)                        # (after bind: a batch-size change there re-allocates its own set)
        rt.backward(x, ctx.B, dout.contiguous(), rt.dims[-1], None)
        return (None, None, None) + tuple(grads)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, num_blocks, output_dim=None, dropout=0., **kwargs):
        super().__init__()
        self.output_dim = output_dim
        self.dropout_p = float(dropout)
        layers = []
        for i in range(num_blocks):
            layers += [nn.Linear(input_dim if i == 0 else hidden_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout)]
        if output_dim is not None:
            layers.append(nn.Linear(hidden_dim, output_dim))
        self.module_list = nn.ModuleList(layers)
        self._dims = [input_dim] + [hidden_dim] * num_blocks + ([output_dim] if output_dim is not None else [])
        self._rt: Optional[MlpRuntime] = None
        self._drop_step = 0
        self._site_base = 1 << 20 | (16 * next(_site_counter))

    def _bump_step(self) -> int:
        self._drop_step += 1
        return self._drop_step

    def _step_args(self, device):
        """(step, step_dev): see modules.mixer._HipTower._step_args."""
        if not config.device_dropout_step():
            return self._bump_step(), None
        ctr = getattr(self, "_drop_counter", None)
        if ctr is None or ctr.device != device:
            ctr = self._drop_counter = torch.full((1,), self._drop_step, dtype=torch.int32, device=device)
        L.check(L.lib().m2m_counter_add(ctr.data_ptr(), 1, L.stream_ptr()), "counter_add")
        self._drop_step += 1
        return 0, ctr

    def _linears(self):
        return [m for m in self.module_list if isinstance(m, nn.Linear)]

    def _linear_params(self):
        return [p for lin in self._linears() for p in (lin.weight, lin.bias)]

    def _param_pairs(self):
        return [(lin.weight, lin.bias) for lin in self._linears()]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("m2_mixer_amd modules run on the GPU only (MI355X); there is no CPU path")
        if x.dim() != 2 or x.shape[1] != self._dims[0]:
            raise RuntimeError(f"expected (B, {self._dims[0]}), got {tuple(x.shape)}")
        if self._rt is None:
            self._rt = MlpRuntime(self._dims, self.output_dim is not None, self.dropout_p, self._site_base)
        self._rt.bind(self._param_pairs(), None, x.shape[0])
        return _MlpFunction.apply(x.contiguous().float(), self, self.training, *self._linear_params())
