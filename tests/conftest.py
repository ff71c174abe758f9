import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference in oracle/_ref")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the HIP library and the oracle port are built (no-op when fresh)."""
    import __graft_entry__

    __graft_entry__.build()
