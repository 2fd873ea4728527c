"""Fusion callables with the reference's shape algebra (reference: modules/fusion.py).

ConcatFusion is the one every BASELINE config uses (cfg/avmnist/avmnist_m2-mixer_B.yml:52); in the
fused training engine it costs nothing (both towers write straight into the halves of one buffer,
see engine.py); as a stand-alone callable it is a single torch.cat.  The other parameter-free
fusions are kept for API parity with the reference's registry and its shape tests.
"""
from __future__ import annotations

import math

import torch
from torch import nn


def _dim_query_guard(args, dim):
    if dim is not None and not isinstance(args[0], int):
        raise ValueError("The dim argument is only used if the first argument is an int.")


def _same_shape_rule(args, dim):
    _dim_query_guard(args, dim)
    if args[0] != args[1]:
        raise ValueError("Input shapes must be equal")
    return args[0]


class ConcatFusion:
    """torch.cat along `dim` (reference: modules/fusion.py:112-146)."""

    def __init__(self, dim=1, **kwargs):
        self.dim = dim

    def __call__(self, *tensors):
        return torch.cat(tensors, dim=self.dim)

    def get_output_shape(self, *args, dim=None):
        _dim_query_guard(args, dim)
        if dim is not None:
            return sum(args) if dim == self.dim else args[0]
        out = list(args[0])
        out[self.dim] = sum(a[self.dim] for a in args)
        return tuple(out)


class ConcatDynaFusion:
    """cat on dim 1, then the result doubled on dim 2 (reference: modules/fusion.py:149-187)."""

    def __init__(self, dim=1, **kwargs):
        self.dim = dim

    def __call__(self, *tensors):
        a = torch.cat(tensors, dim=1)
        return torch.cat([a, a], dim=2)

    def get_output_shape(self, *args, dim=None):
        _dim_query_guard(args, dim)
        if dim is not None:
            return (int(math.sqrt(args[0])) * 2) ** 2 if dim == self.dim else args[0]
        out = list(args[0])
        for a in args[1:]:
            out[1] += a[1]
            out[2] += a[2]
        return tuple(out)


class MaxFusion:
    def __init__(self, **kwargs):
        pass

    def __call__(self, *tensors):
        return torch.maximum(*tensors)

    @staticmethod
    def get_output_shape(*args, dim=None):
        return _same_shape_rule(args, dim)


class SumFusion:
    def __init__(self, **kwargs):
        pass

    def __call__(self, *tensors):
        return torch.add(*tensors)

    @staticmethod
    def get_output_shape(*args, dim=None, **kwargs):
        return _same_shape_rule(args, dim)


class MeanFusion:
    def __init__(self, **kwargs):
        pass

    def __call__(self, *tensors):
        return torch.stack(tensors).mean(0)

    @staticmethod
    def get_output_shape(*args, dim=None, **kwargs):
        return _same_shape_rule(args, dim)


class ExtraConcatFusion:
    """Stack the modalities on a new axis `dim` (reference: modules/fusion.py:224-255)."""

    def __init__(self, dim=1, **kwargs):
        self.dim = dim

    def __call__(self, *tensors):
        return torch.stack(tensors, dim=self.dim)

    def get_output_shape(self, *args, dim=None, num_modality=2):
        _dim_query_guard(args, dim)
        if dim is not None and dim == self.dim:
            return args[0]
        out = list(args[0])
        out.insert(self.dim, num_modality)
        return tuple(out)


class BiModalGatedUnit(nn.Module):
    """z * tanh(W1 a) + (1 - z) * tanh(W2 b), z = sigmoid(Wz [a, b]) (reference: modules/fusion.py:7-55)."""

    def __init__(self, mod1_in, mod2_in, out_size, **kwargs):
        super().__init__()
        self.out_size = out_size
        self.mod1_hidden = nn.Linear(mod1_in, out_size)
        self.mod2_hidden = nn.Linear(mod2_in, out_size)
        self.z_hidden = nn.Linear(mod1_in + mod2_in, out_size)

    def forward(self, mod1, mod2):
        z = torch.sigmoid(self.z_hidden(torch.cat([mod1, mod2], dim=-1)))
        return z * torch.tanh(self.mod1_hidden(mod1)) + (1 - z) * torch.tanh(self.mod2_hidden(mod2))

    def get_output_shape(self, *args, dim=None):
        _dim_query_guard(args, dim)
        if dim is not None:
            return self.out_size if dim == -1 else args[0]
        out = list(args[0])
        out[-1] = self.out_size
        return tuple(out)
