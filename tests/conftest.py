import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Everything the tests load is built in-tree once per session."""
    import __graft_entry__
    __graft_entry__.build()


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
