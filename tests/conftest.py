import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nf_lib():
    """libnf_hip.so, built if absent (hipcc cross-compiles without a GPU)."""
    from normalizing_flows_amd import _lib
    if not os.path.exists(_lib.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()
