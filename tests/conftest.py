import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never be silently skipped on the GPU box: they are only
    deselected by -m "not gpu"; if selected without a device they fail."""
    return


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    assert torch.cuda.is_available(), "this test needs a GPU (select with -m gpu on the GPU box)"
    return torch.device("cuda:0")
