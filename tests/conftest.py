import os
import sys

import pytest

# torch bundles its own libamdhip64.so.7; libfgo.so links the system one with the same SONAME.  Whichever is loaded
# first serves both, and torch only works on its own copy, so any process that uses torch AND libfgo (the multi-GPU
# tests, bench.py) must import torch first.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure only)."""
    from tests import orc_binding
    return orc_binding
