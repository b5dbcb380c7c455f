import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes tens of seconds on the CPU")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One device context for the whole GPU session (fails loudly if the HIP library is missing)."""
    from neklab_amd import host
    ctx = host.Context(0)
    yield ctx
