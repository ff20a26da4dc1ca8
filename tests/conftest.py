import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = "3d-reconstruction-triangulation_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def load_pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def tdr():
    """The product package (ctypes binding of libtdr_amd.so)."""
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure only)."""
    import tdr_oracle
    tdr_oracle.lib()
    return tdr_oracle
