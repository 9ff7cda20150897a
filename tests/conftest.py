import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))   # tests may use the oracle as the checker

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a fresh checkout has no libspcube_hip.so (built artefacts are not in the history): build it once
    # (hipcc cross-compiles for gfx950 without a GPU) so that the ABI / export tests have something to load
    lib = os.path.join(REPO, "spectral_cube_amd", "libspcube_hip.so")
    if not os.path.exists(lib):
        import shutil
        if shutil.which(os.environ.get("HIPCC", "hipcc")):
            import importlib
            importlib.import_module("__graft_entry__").build()


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def assert_close(a, b, rtol=0.0, atol=0.0, what=""):
    """allclose with exact NaN / inf pattern match."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    nan_diff = np.isnan(a) != np.isnan(b)
    assert not nan_diff.any(), what + ": NaN pattern differs (%d vs %d NaN), first at %s" % (
        np.isnan(a).sum(), np.isnan(b).sum(), np.argwhere(nan_diff)[:4].tolist() + np.argwhere(nan_diff)[-2:].tolist())
    fin = np.isfinite(a) & np.isfinite(b)
    err = np.abs(a[fin] - b[fin]) - (atol + rtol * np.abs(b[fin]))
    assert err.size == 0 or err.max() <= 0, "%s: max excess error %.3e" % (what, err.max())
    inf = ~fin & ~np.isnan(a)
    assert np.array_equal(a[inf], b[inf]), what + ": inf pattern differs"


@pytest.fixture(scope="session")
def gpu():
    from spectral_cube_amd import _lib
    _lib.require_gpu()
    return 0
