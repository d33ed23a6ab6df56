import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~30 s on 8 CPU cores")


@pytest.fixture(scope="session")
def lib():
    from stabletriton_amd import _C
    return _C.load()


@pytest.fixture(scope="session")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests selected but no GPU is visible"
    return torch.device("cuda:0")
