import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): C restatement + numpy restatements of the reference."""
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def ivxlib():
    """The product library.  GPU tests must exercise the HIP path: no device -> hard failure, never a skip."""
    from invesalius3_amd import _lib

    _lib.lib()
    _lib.require_device()
    return _lib


def synth_volume(shape, seed=20260924, dtype=np.int16):
    """Small cousin of the V512 generator: blobs + low-frequency wave + noise, HU-like range."""
    rng = np.random.default_rng(seed)
    dz, dy, dx = shape
    z, y, x = np.meshgrid(np.linspace(0, 1, dz), np.linspace(0, 1, dy), np.linspace(0, 1, dx), indexing="ij")
    f = np.zeros(shape, np.float64)
    for _ in range(6):
        c = rng.uniform(0.15, 0.85, 3)
        s = rng.uniform(0.08, 0.25)
        f += 1800.0 * np.exp(-((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / (2 * s * s))
    f += 150.0 * np.sin(6.0 * x + 2.0 * y) * np.cos(5.0 * z)
    f += rng.normal(0, 25, shape)
    return np.clip(f - 1000.0, -1024, 3071).astype(dtype)
