import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure the in-tree HIP library exists and is current (no-op when it is; hipcc cross-compiles without a GPU)
    from cspn_monodepth_amd import _lib
    _lib.build()


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def rel_err(got, want, floor=1e-6):
    """max |got-want| / max(|want|, floor) over finite entries; inf if the NaN patterns differ."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        return float("inf")
    fin = np.isfinite(got) & np.isfinite(want)
    if not fin.any():
        return 0.0
    return float((np.abs(got - want)[fin] / np.maximum(np.abs(want[fin]), floor)).max())


def rmse(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    fin = np.isfinite(got) & np.isfinite(want)
    return float(np.sqrt(((got - want)[fin] ** 2).mean())) if fin.any() else 0.0


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as c
    c.build()
    return c
