import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def libpath():
    """libairfe.so, built in-tree with hipcc if it is not there yet (cross-compiles without a GPU)."""
    from airslam_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.LIB_PATH


def gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
