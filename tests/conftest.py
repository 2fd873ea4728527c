import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible and -m gpu was not asked for."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---- observed maxima of the bf16 parity errors (printed at the end of a run: tolerances follow what is observed) -------------
OBSERVED = {}


def observe(kind: str, value: float, tol: float) -> float:
    """Record the largest error of a kind seen in this session (and the tolerance it is held to); returns the value."""
    cur = OBSERVED.get(kind)
    if cur is None or value > cur[0]:
        OBSERVED[kind] = (float(value), float(tol))
    return value


def pytest_terminal_summary(terminalreporter):
    if OBSERVED:
        terminalreporter.write_line("observed parity errors (max over the session | tolerance):")
        for k in sorted(OBSERVED):
            v, t = OBSERVED[k]
            terminalreporter.write_line(f"  {k:46s} {v:.3e} | {t:.1e}")
