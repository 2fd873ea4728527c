"""Helpers to compare tensors with the digests stored in tests/golden/*.npz."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def sample_index(numel):
    return np.linspace(0, numel - 1, 257).astype(np.int64)


def check(store, name, t: torch.Tensor, atol, rtol=0.0, what=""):
    """Compare `t` with digest `name` (full tensor or 257-point sample + l2 norm)."""
    t = t.detach().to("cpu", torch.float64).flatten()
    numel = int(store[f"{name}//numel"])
    assert t.numel() == numel, f"{what}{name}: numel {t.numel()} != {numel}"
    if f"{name}//full" in store:
        ref = torch.from_numpy(store[f"{name}//full"].astype(np.float64))
        got = t
    else:
        ref = torch.from_numpy(store[f"{name}//sample"].astype(np.float64))
        got = t[torch.from_numpy(sample_index(numel))]
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bool(bad.any()), (f"{what}{name}: max abs err {err.max().item():.3e} "
                                 f"(tol {atol:g}+{rtol:g}*|ref|), ref max {ref.abs().max().item():.3e}")
    l2 = float(store[f"{name}//l2"])
    got_l2 = float(t.norm())
    assert abs(got_l2 - l2) <= (atol * np.sqrt(numel) + (rtol + 1e-6) * l2 + 1e-12), \
        f"{what}{name}: l2 {got_l2} vs {l2}"
    return float(err.max())
