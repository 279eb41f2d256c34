import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the HIP library and the oracle are built (no-op when the .so files are current)."""
    import __graft_entry__ as ge

    ge.build()
