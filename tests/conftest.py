import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the `gpu` tests instead of erroring in their fixtures."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (run with `pytest -m gpu` via gpurun)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_small():
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "small_T4.pt"), weights_only=False)


@pytest.fixture(scope="session")
def golden_small_v21():
    """VideoLLaMA2.1-shaped fixture (SigLIP tower + stc_connector_v35 + Qwen2), minted from the real reference."""
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "small_v21_T4.pt"), weights_only=False)
