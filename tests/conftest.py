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
def rng():
    return np.random.default_rng(20260723)


@pytest.fixture(scope="session")
def hip():
    """The HIP library on a box with a device; GPU tests fail loudly if it is unusable."""
    from smmregrid_amd import _lib
    _lib.load()
    n = _lib.device_count()
    assert n > 0, "no HIP device visible: GPU tests cannot run (no CPU fallback exists)"
    return _lib
