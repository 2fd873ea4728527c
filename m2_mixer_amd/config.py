"""Process-wide knobs of the HIP path."""
from __future__ import annotations

import os

from . import _lib as L

_state = {
    # "bf16": bf16 MFMA operands, fp32 accumulate / residual / LayerNorm  (the training + bench mode)
    # "fp32": exact fp32 MFMA (parity mode: tracks the reference CPU path to ~1e-5)
    "precision": os.environ.get("M2M_PRECISION", "bf16"),
    "seed": None,
}


def set_precision(name: str) -> None:
    if name not in L.PREC_BY_NAME:
        raise ValueError(f"precision must be one of {sorted(L.PREC_BY_NAME)}")
    _state["precision"] = name


def get_precision() -> str:
    return _state["precision"]


def prec_id(name: str | None = None) -> int:
    return L.PREC_BY_NAME[name or _state["precision"]]


def set_dropout_seed(seed: int) -> None:
    _state["seed"] = int(seed) & 0xFFFFFFFF


def dropout_seed() -> int:
    if _state["seed"] is None:
        import torch
        _state["seed"] = int(torch.initial_seed()) & 0xFFFFFFFF
    return _state["seed"]
