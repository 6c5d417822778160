import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes minutes (full-size runs); still part of -m gpu")


@pytest.fixture(scope="session")
def ctx():
    from femo_amd import _lib
    from femo_amd.engine import Context
    if _lib.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests need a GPU (femo_amd has no CPU fallback)")
    c = Context(0)
    yield c
