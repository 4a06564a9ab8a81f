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


def bits_equal(a, b, **ctx):
    """torch.equal(a, b) — and on a mismatch a dump of WHICH elements differ under gpurun_out/mismatch/ (VERDICT r4 next #1a: the
    one bit mismatch round 4 ever saw left nothing to analyse).  The dump holds the differing index set (per dimension, the first
    elements with both values), the host protocol's state (flag sequence number, workspace pointers, fallbacks) and, when
    the tensors are [B,(1,)H,W] planes of a shape with a resident tiling, the tile / row-in-tile / column-in-tile of every differing
    pixel of the first image concerned."""
    import json
    import torch
    if a.shape == b.shape and torch.equal(a, b):
        return True
    try:
        from cspn_monodepth_amd import functional as F
        rec = dict(test=os.environ.get("PYTEST_CURRENT_TEST", "?"), shape_a=list(a.shape), shape_b=list(b.shape), dtype=str(a.dtype),
                   ctx={k: (v if isinstance(v, (int, float, str, bool, type(None), list, tuple, dict)) else repr(v)) for k, v in ctx.items()},
                   fallbacks=F.resident_fallbacks(), resident_mode=F._RESIDENT_MODE)
        for idx, st in F._RES.items():
            rec["device%d" % idx] = dict(seq=st["seq"], host_err=[int(v) for v in st["host_err_np"]], dirty=st["dirty"],
                                         journal=len(st["journal"]), last_seq=st.get("last_seq"),
                                         workspaces={str(k): hex(w.data_ptr()) for k, w in st["work"].items()})
        if a.shape == b.shape:
            af, bf = a.detach(), b.detach()
            diff = ~((af == bf) | (torch.isnan(af) & torch.isnan(bf))) if af.is_floating_point() else (af != bf)
            only_nan_vs_nan = bool((torch.isnan(af) & torch.isnan(bf)).any()) if af.is_floating_point() else False
            idx = diff.nonzero().cpu().numpy()
            rec.update(n_diff=int(idx.shape[0]), nan_in_a=int(torch.isnan(af).sum()) if af.is_floating_point() else 0,
                       nan_in_b=int(torch.isnan(bf).sum()) if bf.is_floating_point() else 0, nan_at_same_places=only_nan_vs_nan)
            if idx.shape[0]:
                per_dim = []
                for d in range(idx.shape[1]):
                    u = np.unique(idx[:, d])
                    per_dim.append(dict(min=int(u.min()), max=int(u.max()), count=int(u.size), values=[int(v) for v in u[:96]]))
                rec["per_dim"] = per_dim
                ac, bc = af.cpu().numpy(), bf.cpu().numpy()
                rec["first"] = [dict(index=[int(v) for v in i], a=repr(ac[tuple(i)]), b=repr(bc[tuple(i)])) for i in idx[:48]]
                if a.dim() in (3, 4) and "T" in ctx:
                    B, H, W = a.shape[0], a.shape[-2], a.shape[-1]
                    rp = F.resident_plan(B, H, W, int(ctx["T"]), int(bool(ctx.get("sparse"))), 256)
                    if rp is not None:
                        rec["resident_plan"] = {k: rp[k] for k in ("steps_per_phase", "tiles_x", "tiles_y", "tile_w", "tile_h", "quads_per_thread", "images_per_launch", "launches")}
                        img = int(idx[0, 0])
                        sel = idx[idx[:, 0] == img]
                        ys, xs = sel[:, -2], sel[:, -1]
                        tiles = {}
                        for y, x in zip(ys[:20000], xs[:20000]):
                            key = "%d,%d" % (y // rp["tile_h"], x // rp["tile_w"])
                            t = tiles.setdefault(key, dict(n=0, row_in_tile=[10 ** 9, -1], col_in_tile=[10 ** 9, -1]))
                            t["n"] += 1
                            ry, rx = int(y % rp["tile_h"]), int(x % rp["tile_w"])
                            t["row_in_tile"] = [min(t["row_in_tile"][0], ry), max(t["row_in_tile"][1], ry)]
                            t["col_in_tile"] = [min(t["col_in_tile"][0], rx), max(t["col_in_tile"][1], rx)]
                        rec["image"] = img
                        rec["launch_of_image"] = img // max(rp["images_per_launch"], 1)
                        rec["tiles_ty_tx"] = tiles
        out_dir = os.path.join(ROOT, "gpurun_out", "mismatch")
        os.makedirs(out_dir, exist_ok=True)
        name = "".join(c if c.isalnum() or c in "-_." else "_" for c in rec["test"])[-150:]
        n = len(glob.glob(os.path.join(out_dir, name + "*.json")))
        with open(os.path.join(out_dir, "%s.%d.json" % (name, n)), "w") as fh:
            json.dump(rec, fh, indent=1)
        print("bits_equal: MISMATCH dumped to gpurun_out/mismatch/%s.%d.json (%s differing)" % (name, n, rec.get("n_diff", "shape")))
    except Exception as exc:       # noqa: BLE001 — the dump must never hide the assertion it serves
        print("bits_equal: mismatch (dump failed: %r)" % (exc,))
    return False


class lds_poison(object):
    """with lds_poison(): every kernel the engine launches is preceded by a NaN fill of the whole LDS of every CU
    (include/cspn_hip.h: cspn_debug_set_lds_poison) — a kernel that reads LDS it never wrote then shows it in its result."""

    def __init__(self, pattern=0x7fc00000):
        self.pattern = pattern

    def __enter__(self):
        import ctypes
        from cspn_monodepth_amd import _lib
        self.prev = ctypes.c_int(0)
        _lib.check(_lib.lib().cspn_debug_set_lds_poison(1, self.pattern, ctypes.byref(self.prev)), "cspn_debug_set_lds_poison")

    def __exit__(self, *exc):
        from cspn_monodepth_amd import _lib
        _lib.lib().cspn_debug_set_lds_poison(self.prev.value, self.pattern, None)
        return False
