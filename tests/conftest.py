import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    """PyTorch bundles its own HIP runtime: in a process that uses both, torch has to initialise first (tests that hand
    device tensors to the library need it; harmless without a GPU)."""
    try:
        import torch

        torch.cuda.is_available()
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.loader import Oracle, build_oracle

    build_oracle()
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    from tests.golden_util import load_golden

    return load_golden()
