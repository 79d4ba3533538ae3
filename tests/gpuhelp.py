"""Shared helpers for the -m gpu tests: the product binding, a context fixture, error metrics."""
import os
import sys

import numpy as np
import pytest

# torch ships its own ROCm runtime; a process that uses both must load torch FIRST so that libmsdr.so binds
# to the runtime torch already initialised (two HIP runtimes cannot both claim the GPU).  Tests that pass torch
# buffers / streams to the library rely on this order.
try:
    import torch  # noqa: F401
except ImportError:      # torch is optional for the GPU tests that use the library's own allocator
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimal-sdr_amd", "python"))
import msdr  # noqa: E402,F401


@pytest.fixture(scope="module")
def ctx():
    c = msdr.Context(0)          # raises (never falls back) if the HIP library / a gfx950 GPU is missing
    yield c
    c.close()


def rel_rms(got, want):
    """||got - want||_2 / ||want||_2 (the north-star's error measure)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.sqrt(((got - want) ** 2).sum() / max((want ** 2).sum(), 1e-300)))
