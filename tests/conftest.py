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


@pytest.fixture()
def in_tmp(tmp_path, monkeypatch):
    """Run inside a scratch dir that has the repo's cost/slo files, with an empty local registry."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("cost.yaml", "slo.json"):
        (tmp_path / name).write_text(open(os.path.join(root, name)).read())
    os.symlink(os.path.join(root, "runners"), tmp_path / "runners")
    os.symlink(os.path.join(root, "profiles"), tmp_path / "profiles")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("KVMINI_LOCAL_REGISTRY", str(tmp_path / "registry.json"))
    monkeypatch.setenv("KVMINI_CLUSTER_BACKEND", "fake")
    return tmp_path
