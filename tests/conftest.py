import os
import sys

import pytest

# The tests run the matcher in every form (environment switches read at each launch): the library honours those only in
# a process that asked for it before its first call (nhip_common.h, tunable()).
os.environ.setdefault("NHIP_TUNABLES", "1")

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


@pytest.fixture
def chi_square_cases(n=4000, seed=11):
    """(poses, pair_src, pair_tgt, cov) for LCMatcher's chi-square test (tests/test_hostside.py, tests/test_lc_gpu.py)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    poses = np.concatenate([rng.uniform(-40, 40, (300, 2)), rng.uniform(-3.2, 3.2, (300, 1))], axis=1)
    src, tgt = rng.integers(0, 300, n).astype(np.int32), rng.integers(0, 300, n).astype(np.int32)
    src[:8] = tgt[:8]                                             # a scan is never its own match
    # symmetric positive definite blocks of very different scale, a few nearly singular, a few not symmetric
    a = rng.normal(size=(n, 2, 2)) * (10.0 ** rng.uniform(-3, 0.5, (n, 1, 1)))
    cov = (a @ a.transpose(0, 2, 1) + 1e-9 * np.eye(2)).astype(np.float32)
    cov[8:40, 0, 1] *= 1.01
    cov[40:48] = np.array([[1.0, 2.0], [2.0, 4.0]], np.float32)    # det == 0: inf / NaN as cov.inverse() gives
    cov[48:56] = 0.0
    return poses, src, tgt, cov
