import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def sd0():
    from pope_amd.synth import synthetic_state_dict
    return synthetic_state_dict(seed=0)


@pytest.fixture(scope="session")
def hip_lib():
    """Built + loaded C-ABI library (build is a no-op when the .so is current)."""
    from pope_amd import _lib
    _lib.build()
    return _lib.lib()
