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


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the HIP library / host tools / oracle once if a fresh checkout has none of them
    (hipcc cross-compiles without a GPU; ~1 minute).  Never rebuilds an existing artefact."""
    pkg_dir = os.path.join(ROOT, PKG)
    needed = [os.path.join(pkg_dir, n) for n in ("libtdr_amd.so", "host_demo", "tdr_main", "host_io_check")]
    needed.append(os.path.join(ROOT, "oracle", "libtdr_oracle.so"))
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


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
