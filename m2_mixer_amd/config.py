"""Process-wide knobs of the HIP path."""
from __future__ import annotations

import os

from . import _lib as L

_state = {
    # "bf16": bf16 MFMA operands, fp32 accumulate / residual / LayerNorm  (the training + bench mode)
    # "fp32": exact fp32 MFMA (parity mode: tracks the reference CPU path to ~1e-5)
    "precision": os.environ.get("M2M_PRECISION", "bf16"),
    "seed": None,
    # dropout step counters of the MODULE path on the device (a captured hipGraph must not bake the step in: graphs.GraphedStep)
    "device_step": False,
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


def set_device_dropout_step(on: bool) -> None:
    """Module path: keep every tower's / MLP's dropout step counter in device memory (advanced by a tiny launch in front of the
    forward, read by the kernels) instead of passing a host integer.  Needed when a training step is captured into a hipGraph
    (m2_mixer_amd.graphs.GraphedStep turns it on): a host integer would be baked into the graph and every replay would draw the
    SAME dropout masks.  The mask stream is the same in both modes (step 1, 2, 3, ...).  Not re-entrant: a backward must run
    before the next forward of the same module (always true for shared_step -> backward -> optimizer.step)."""
    _state["device_step"] = bool(on)


def device_dropout_step() -> bool:
    return _state["device_step"]
