import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def B():
    import bdf_amd
    return bdf_amd


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx(B):
    """one device context for the GPU tests (seed 1234)"""
    from bdf_amd import Context
    c = Context(seed=1234)
    yield c
    c.close()
