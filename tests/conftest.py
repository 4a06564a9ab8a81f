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


def pytest_collection_modifyitems(config, items):
    # a box without a ROCm GPU reports the `gpu` tests as skipped instead of failing them one by one
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:       # noqa: BLE001
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (MI355X): run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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


@pytest.fixture(autouse=True)
def _fresh_resident_fallback_state():
    """Time-out tests count fallbacks and may switch mode "auto" off for the process (functional._note_fallback): every test
    starts from a clean count and leaves the mode as it found it."""
    from cspn_monodepth_amd import functional as F
    mode = F._RESIDENT_MODE
    F._FALLBACKS, F._FALLBACK_WARNED = 0, False
    yield
    F._RESIDENT_MODE = mode
    F._FALLBACKS, F._FALLBACK_WARNED = 0, False


def occupy_lib():
    """tests/support/libcspn_occupy.so (a co-tenant kernel for the contention tests), built on demand with hipcc."""
    import ctypes
    import shutil
    import subprocess
    src = os.path.join(ROOT, "tests", "support", "occupy.hip")
    so = os.path.join(ROOT, "tests", "support", "libcspn_occupy.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-Wno-unused-value", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
    lib.occupy.restype = ctypes.c_int
    return lib
