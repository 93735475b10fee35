import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu on the GPU box")
    config.addinivalue_line("markers", "slow: long-running")


@pytest.fixture(autouse=True)
def _seed_and_dtype():
    # same hygiene as the reference's tests/conftest.py:18,27-29,50-52
    torch.manual_seed(1)
    torch.set_default_dtype(torch.float32)
    yield


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
