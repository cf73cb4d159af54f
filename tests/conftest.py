import os
import sys

import pytest

# The train engine's workspace sets are views of one arena and alias each other across batch shapes (engine._Arena).  Under the tests every
# activation of a set first fills its whole extent with NaN: a kernel that relied on zero-initialised or stale workspace memory fails a
# parity test loudly instead of passing by luck.
os.environ.setdefault("MSTTS_ARENA_POISON", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
