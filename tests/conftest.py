import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); deselect with -m 'not gpu'")
    config.addinivalue_line("markers", "slow: long-running")


def _gpu_ok() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_ok():
        return
    skip = pytest.mark.skip(reason="no ROCm GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture()
def mock_server():
    from kvmini.server.mock import MockConfig, start_mock_in_thread

    mock, url, stop = start_mock_in_thread(MockConfig(ttft_ms=3.0, tpot_ms=0.2, jitter=0.05))
    yield mock, url
    stop()
