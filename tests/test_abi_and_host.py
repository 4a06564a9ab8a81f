"""CPU tests of the boundary: the C-ABI library loads and exports every symbol the header declares,
the host-side mirror of the reference interface behaves like the reference's module API, and nothing
in the product path can silently fall back to a CPU implementation."""
import ctypes
import os
import re

import pytest
import torch

import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import _lib
from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "cspn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cspn\w*)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    lib = ctypes.CDLL(_lib.SO_PATH)
    declared = _header_functions()
    assert set(declared) == set(_lib.EXPORTS), (declared, _lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.lib().cspn_abi_version() == _lib.ABI_VERSION == 10
    assert _lib.lib().cspn_propagate_workspace_bytes(24, 228, 304, 24, _lib.CSPN_F32, 0) == 2 * 24 * 228 * 304 * 4
    assert _lib.lib().cspn_propagate_workspace_bytes(24, 228, 304, 24, _lib.CSPN_F32, 1) == 0


def test_module_api_matches_reference_contract():
    m = pkg.CSPN_new.AffinityPropagate(24, 3)            # unet_cspn_nyu.py:357-358
    assert (m.prop_time, m.prop_kernel, m.in_feature, m.out_feature) == (24, 3, 1, 1)
    assert len(m.state_dict()) == 0 and len(list(m.parameters())) == 0
    with pytest.raises(ValueError):
        pkg.CSPN_new.AffinityPropagate(24, 5)
    p = pkg.CSPN_ours.AffinityPropagate(prop_time=24)    # unet_ours.py:304-305
    assert p.times == 24 and len(p.state_dict()) == 0
    import inspect
    assert list(inspect.signature(m.forward).parameters) == ["guidance", "blur_depth", "sparse_depth"]
    assert list(inspect.signature(p.forward).parameters) == ["x", "guided", "sparse_depth"]


def test_no_cpu_fallback():
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    with pytest.raises(RuntimeError, match="ROCm device"):
        m(torch.randn(1, 8, 4, 4), torch.rand(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match="ROCm device"):
        pkg.CSPN_ours.AffinityPropagate(3)(torch.rand(1, 1, 4, 4), torch.randn(1, 8, 4, 4))
    with pytest.raises(RuntimeError):
        pkg.evaluation.metric_sums(torch.rand(4), torch.rand(4))
    # the product package never imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "cspn_monodepth_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_shard_bounds_and_finalize():
    ev = pkg.evaluation
    for n, w in ((24, 8), (8, 8), (24, 5), (3, 4)):
        spans = [ev.shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    import numpy as np
    from oracle import cspn_oracle as orc
    from conftest import load_golden
    z = load_golden("g7_metrics")
    fin = ev.finalize_metrics(orc.metric_sums(z["pred"], z["target"]))
    for k, v in zip(ev.METRIC_NAMES, z["metrics"]):
        assert np.isclose(fin[k], v, rtol=2e-6), k


def test_widening_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation of the PAC-conv / un-pooling entries happens on the host before any launch: a failing call
    returns 0 and leaves a message in cspn_last_error() (the In-Place ABN convention the header adopts)."""
    L = _lib.lib()
    G = _lib.cspn_conv_geometry
    err = lambda: L.cspn_last_error().decode()                                     # noqa: E731
    ok_geom = G(3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0)
    one = ctypes.c_void_p(16)                                                      # never dereferenced: validation fails first
    assert L.cspn_pac_conv2d(one, one, one, _lib.CSPN_F32, 1, 3, 2, 8, 8, ctypes.byref(ok_geom), None) == 0
    assert "Incompatible input and kernel sizes" in err()                          # kernel_ch must be 1 or C (pac.py:77-78)
    assert L.cspn_pac_conv2d(one, one, one, 7, 1, 3, 1, 8, 8, ctypes.byref(ok_geom), None) == 0 and "dtype" in err()
    assert L.cspn_pac_conv2d(None, one, one, _lib.CSPN_F32, 1, 3, 1, 8, 8, ctypes.byref(ok_geom), None) == 0 and "null" in err()
    assert L.cspn_pac_conv2d(one, one, one, _lib.CSPN_F32, 1, 3, 1, 8, 8, None, None) == 0 and "geom" in err()
    big = G(9, 9, 1, 1, 0, 0, 1, 1, 0, 0, 0)                                       # window larger than the padded input
    assert L.cspn_pac_conv2d(one, one, one, _lib.CSPN_F32, 1, 1, 1, 4, 4, ctypes.byref(big), None) == 0 and "empty output" in err()
    tr = G(3, 3, 2, 2, 1, 1, 1, 1, 1, 1, 1)                                        # transposed geometry is nd2col-only
    assert L.cspn_pac_conv2d_grad_input(one, one, one, _lib.CSPN_F32, 1, 1, 1, 4, 4, ctypes.byref(tr), None) == 0
    assert "nd2col only" in err()
    ho, wo = ctypes.c_int(), ctypes.c_int()
    assert L.cspn_pac_out_size(6, 7, ctypes.byref(tr), ctypes.byref(ho), ctypes.byref(wo)) == 1 and (ho.value, wo.value) == (12, 14)
    assert L.cspn_unpool2d(one, one, _lib.CSPN_F32, 4, 5, 6, 2, 11, 12, None) == 0 and "must lie in" in err()   # oH > 2*H
    assert L.cspn_unpool2d(one, one, _lib.CSPN_F32, 4, 5, 6, 0, 5, 6, None) == 0 and "scale" in err()
    assert L.cspn_unpool2d_backward(one, None, _lib.CSPN_F32, 4, 5, 6, 2, 10, 12, None) == 0 and "null" in err()


def test_batch_average_meter_reproduces_reference_averaging():
    """the reference's eval prints the batch-weighted mean of per-batch metrics (Result.evaluate +
    AverageMeter, libs/metrics.py:49-127), not the pixel-weighted global figure.  Checked against the numpy oracle's
    restatement of Result.evaluate (pinned by golden G7) on batches with unequal valid-pixel counts."""
    import numpy as np
    import torch
    from cspn_monodepth_amd import evaluation as ev
    from oracle import cspn_oracle as orc
    rng = np.random.default_rng(3)
    meter = ev.BatchAverageMeter()
    want = np.zeros(10)
    total = np.zeros(10)
    n_tot = 0
    for n, frac in ((1, 0.9), (3, 0.4), (2, 0.7)):
        t = rng.uniform(0.5, 10, (n, 1, 12, 16)).astype(np.float32)
        t[rng.random(t.shape) > frac] = 0
        p = (np.abs(t + rng.normal(0, 0.3, t.shape)) + 0.05).astype(np.float32)
        sums = orc.metric_sums(p, t)
        meter.update(torch.from_numpy(sums), n=n)
        ref, _ = orc.evaluate_metrics(p, t)                       # Result.evaluate of this batch
        want += n * np.asarray(ref)
        total += sums
        n_tot += n
    avg = meter.average()
    assert avg["count"] == n_tot
    for k, v in zip(ev.METRIC_NAMES, want / n_tot):
        assert np.isclose(avg[k], v, rtol=1e-6), k
    glob = ev.finalize_metrics(torch.from_numpy(total))
    assert not np.isclose(glob["rmse"], avg["rmse"], rtol=1e-4)   # the two conventions really differ


def test_resident_plan_host_rules():
    """Host-side tiling rules of the weight-resident launch (no GPU needed: cspn3_resident_plan is pure host code).
    * a launch never has more workgroups than CUs, whole images per launch;
    * phases have an even number of steps whenever there is more than one phase (every phase starts in LDS buffer 0,
      the other buffer being a compile-time distance away), an odd request with several phases has no plan;
    * one depth buffer fits the fixed slot (16352 floats) and the launch fits 160 KB of LDS."""
    from cspn_monodepth_amd import functional as F
    for (B, H, W) in ((24, 228, 304), (8, 352, 1216), (1, 352, 1216), (3, 228, 304), (5, 100, 64)):
        for blend in (0, 1):
            p = F.resident_plan(B, H, W, 24, blend, 256)
            assert p is not None, (B, H, W)
            assert p["tiles_x"] * p["tiles_y"] * p["images_per_launch"] <= 256
            assert p["launches"] == -(-B // p["images_per_launch"])
            assert p["steps_per_phase"] % 2 == 0 and p["lds_bytes"] <= 160 * 1024
            hx = -(-(p["steps_per_phase"] - 1) // 4) * 4
            dr, ls = p["tile_h"] + 2 * (p["steps_per_phase"] - 1) + 2, p["tile_w"] + 2 * hx + 8
            assert dr * ls <= 16352, p
            assert p["quads_per_thread"] <= 5 and p["threads"] == 512
    assert F.resident_plan(24, 228, 304, 24, 0, 256)["steps_per_phase"] == 8            # the fitted cost model's choice
    assert F.resident_plan(8, 352, 1216, 24, 0, 256)["steps_per_phase"] == 8
    assert F.resident_plan(24, 228, 304, 24, 0, 256, steps_per_phase=5) is None         # 5 phases of 5 steps: odd
    assert F.resident_plan(24, 228, 304, 5, 0, 256, steps_per_phase=5) is not None      # a single phase may be odd
    assert F.resident_plan(24, 228, 304, 24, 0, 256, steps_per_phase=6)["steps_per_phase"] == 6
    assert F.resident_plan(4, 64, 30, 24, 0, 256) is None                               # W % 4 != 0


def test_production_shapes_keep_their_resident_instances():
    """The tilings (hence the kernel instances: NQ = quads per thread, CLEAN, phases) the default path runs for the BASELINE
    configs on the 256 CUs of an MI355X — the same table tests/test_hip_production.py asserts on the GPU, checked here without one
    (cspn3_resident_plan / cspnk_resident_plan are pure host code): a change to resident_geometry that moves a production shape to
    another instance fails on the CPU suite already."""
    from cspn_monodepth_amd import functional as F
    keys = ("steps_per_phase", "tiles_x", "tiles_y", "tile_w", "tile_h", "quads_per_thread", "images_per_launch", "launches")
    want = {(24, 228, 304): (8, 2, 5, 152, 46, 5, 24, 1), (3, 228, 304): (8, 7, 12, 44, 19, 1, 3, 1),
            (1, 352, 1216): (8, 11, 22, 112, 16, 2, 1, 1), (8, 352, 1216): (8, 8, 8, 152, 44, 5, 4, 2)}
    for (B, H, W), w in want.items():
        for blend in (0, 1):
            p = F.resident_plan(B, H, W, 24, blend, 256)
            assert tuple(p[k] for k in keys) == w and p["threads"] == 512, ((B, H, W), {k: p[k] for k in keys})
    for blend in (0, 1):            # config 3: K = 5, fp16 guidance — cspnk_d2 with 768 threads, both 12-image halves in one launch
        p = F.kres_plan(5, 24, 228, 304, 12, blend, 256)
        assert tuple(p[k] for k in keys) == (4, 2, 10, 152, 23, 1, 12, 2) and p["threads"] == 768, p


def test_journal_entries_demote_to_weak_references():
    """Host logic of the resident launches' journal (no GPU needed): an old entry holds its tensors weakly — through the tensor
    that owns the storage, so that a view made inside the package can be rebuilt — and reports a dropped output as "nothing to
    repair" and dropped inputs as "cannot be repaired"; version counters tell an in-place overwrite."""
    import gc
    from cspn_monodepth_amd import functional as F
    g = torch.arange(2 * 12 * 4 * 8, dtype=torch.float32).reshape(2, 12, 4, 8)
    d = torch.ones(2, 1, 4, 8)
    d0 = d[:, 0]                                       # a view, as functional._plane makes it
    out = torch.zeros(2, 4, 8)
    e = F._JournalEntry(lambda *a: None, out, (g, d0, None, None), "test")
    assert e.nbytes == (g.numel() + d0.numel() + out.numel()) * 4 and not e.weak
    e.demote()
    assert e.weak
    del d0
    gc.collect()
    o2, ins = e.resolve()
    assert o2.data_ptr() == out.data_ptr() and ins[0].data_ptr() == g.data_ptr()
    assert ins[1].data_ptr() == d.data_ptr() and tuple(ins[1].shape) == (2, 4, 8) and ins[2] is None      # the view, rebuilt from its base
    assert e.untouched(ins)
    g.add_(1.0)
    assert not e.untouched(e.resolve()[1])            # an in-place overwrite shows in the version counter
    del out, o2
    gc.collect()
    assert e.resolve()[0] is None                     # the result was dropped: nobody can read it
    del g, ins
    gc.collect()
    assert e.resolve()[1] is None                     # an input is gone: the call cannot be re-run


def test_journal_entries_accept_inference_tensors_and_training_entries_pin_nothing():
    """ADVICE r5.  (1) Tensors made under torch.inference_mode() have no version counter (reading it raises): a journal entry
    built from them — every unguarded launch builds one, the default scored forward among them — records "cannot verify" instead
    of crashing after the kernel was enqueued, and the repair trusts such inputs.  (2) A training-form entry (redo None) keeps no
    inputs and holds the plane a failed tile poisons weakly from the start."""
    import gc
    from cspn_monodepth_amd import functional as F
    with torch.inference_mode():
        g = torch.ones(2, 12, 4, 8)
        d0 = torch.ones(2, 4, 8)
        out = torch.zeros(2, 4, 8)
        assert g.is_inference()
        e = F._JournalEntry(lambda *a: None, out, (g, d0, None, None), "test")
        assert e.versions[0] is F._NO_VERSION and e.versions[2] is None
        assert e.untouched(e.inputs)
        e.demote()
        o2, ins = e.resolve()
        assert o2.data_ptr() == out.data_ptr() and e.untouched(ins)
    # mixed: an ordinary tensor next to an inference one is still checked
    h = torch.ones(2, 4, 8)
    e2 = F._JournalEntry(lambda *a: None, torch.zeros(1), (g, h), "test")
    h.add_(1.0)
    assert not e2.untouched(e2.inputs)
    # training form: no inputs, weak output, nothing pinned
    hist = torch.zeros(3, 2, 4, 8)
    e3 = F._JournalEntry(None, hist[2], (g, h), "training")
    assert e3.inputs == () and e3.weak and e3.nbytes == 0
    assert e3.out.get().data_ptr() == hist[2].data_ptr()
    del hist
    gc.collect()
    assert e3.out.get() is None


def test_bench_cpu_binding_narrows_and_restores_the_affinity_mask():
    """bench.py --cpu-bind auto: a rank's host threads go to its share of the first allowed cores; the CPU baseline leg gets the
    whole mask back (full_affinity), and the reported host budget is the original one."""
    import importlib.util
    import os
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity on this platform")
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    before = os.sched_getaffinity(0)
    try:
        assert bench.bind_cpus(0, 1, "off") is None and os.sched_getaffinity(0) == before
        mine = bench.bind_cpus(1, 2, "auto")
        if len(before) >= 4:
            k = 4 if len(before) >= 8 else len(before) // 2
            assert mine == sorted(before)[k:2 * k] and os.sched_getaffinity(0) == set(mine)
            assert bench.host_cpu_budget()["sched_affinity"] == len(before)
            with bench.full_affinity():
                assert os.sched_getaffinity(0) == before
            assert os.sched_getaffinity(0) == set(mine)
    finally:
        os.sched_setaffinity(0, before)
