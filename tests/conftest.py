import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import dsa_loader  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dsa():
    return dsa_loader.load()


@pytest.fixture(scope="session")
def oracle(dsa):
    """The CPU oracle (test infrastructure).  Built on demand with g++."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_binding
    return oracle_binding.load(dsa)


@pytest.fixture(scope="session")
def hip(dsa):
    """The HIP product library; fails (not skips) when it is missing or no GPU is usable."""
    import ctypes as C
    b = dsa.product()
    n = C.c_int32()
    b.call("device_count", C.byref(n))
    assert n.value >= 1, "no gfx950 device visible"
    return b
