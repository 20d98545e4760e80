import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TABLE_SEED = 0x5EED7AB1E
SIM_SEED = 0x5EEDCA125


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def cpm():
    """The product package; building the HIP library if needed (hipcc cross-compiles on CPU)."""
    from carparkingmaps_amd import _lib
    _lib.build()
    import carparkingmaps_amd
    return carparkingmaps_amd
