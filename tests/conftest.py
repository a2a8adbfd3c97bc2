import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CUBE_OBJ = os.path.join(GOLDEN, "cube.obj")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(params=["rows", "planes"])
def lbm(request, monkeypatch):
    """The HIP path, once per population layout (include/lbm_abi.h LBM_LAYOUT_*; the library reads
    LBM_LAYOUT at lbm_create).  No fallback: a missing library or GPU fails the test."""
    from cfd_amd import lbm as L
    L.load_library()
    monkeypatch.setenv("LBM_LAYOUT", str(L.LAYOUT_ROWS if request.param == "rows" else L.LAYOUT_PLANES))
    return L
