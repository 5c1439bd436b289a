import importlib
import os
import sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see 3d_adapt_auto_driving_amd/__init__.py (graph replay)

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "3d_adapt_auto_driving_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    return importlib.import_module(PKG_NAME + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def ext():
    """The three drop-in extension modules, imported by the reference's own names."""
    p = pkg()
    if p.DROPIN_DIR not in sys.path:
        sys.path.insert(0, p.DROPIN_DIR)
    import pointnet2_cuda
    import iou3d_cuda
    import roipool3d_cuda

    class E:
        pass
    e = E()
    e.pointnet2, e.iou3d, e.roipool3d = pointnet2_cuda, iou3d_cuda, roipool3d_cuda
    return e
