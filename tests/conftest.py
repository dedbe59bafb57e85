import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (never skip, never fall back) when the HIP path is unavailable."""
    from nautilus_amd import _lib
    _lib.load()
    n = _lib.device_count()
    assert n > 0, "no HIP device visible: the -m gpu tests must run on the GPU box"
    return n


@pytest.fixture(scope="session")
def small_bag():
    from nautilus_amd import synth
    return synth.SynthBag(48)
