"""GPU parity tests: the HIP path (through the C ABI) vs the golden vectors captured from the reference
and vs the C oracle on seeded inputs.  Tolerance (BASELINE.json north_star): 1e-5 relative fp32,
refined-depth RMSE 1e-4."""
import numpy as np
import pytest
import torch

import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import _lib
from conftest import golden_names, load_golden, rel_err, rmse
from oracle import cspn_oracle as orc

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5
RMSE_TOL = 1e-4
DEV = "cuda:0"


def dev(x, dtype=None):
    if x is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t if dtype is None else t.to(dtype)


def run3(g, d, s, T, plan=None):
    m = pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)
    with torch.no_grad():
        out = m(dev(g), dev(d), dev(s))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_native_library_is_loaded():
    assert _lib.lib().cspn_abi_version() == _lib.ABI_VERSION == 10
    maps = open("/proc/self/maps").read()
    assert "libcspn_hip.so" in maps


@pytest.mark.parametrize("name", golden_names("g1_") + golden_names("g2_"))
def test_golden_small_and_degenerate(name):
    z = load_golden(name)
    out = run3(z["guidance"], z["blur"], z.get("sparse"), int(z["T"]))
    assert out.shape == z["out"].shape and out.dtype == np.float32
    assert rel_err(out, z["out"]) <= REL_TOL, name      # includes identical NaN patterns (1x1, zero gates)


@pytest.mark.parametrize("name", golden_names("g3_") + golden_names("g4_"))
def test_golden_full_frames(name, c_oracle):
    z = load_golden(name)
    _, H, W = (int(v) for v in z["shape"])
    ss = int(z["sparse_samples"])
    g, d, s = c_oracle.synthetic_inputs(int(z["seed"]), 1, H, W, 8, None if ss < 0 else ss)
    out = run3(g, d, s, int(z["T"]))
    sub = int(z["sub"])
    assert rel_err(out[:, :, ::sub, ::sub], z["out_sub"]) <= REL_TOL
    assert rmse(out[:, :, ::sub, ::sub], z["out_sub"]) <= RMSE_TOL
    o64 = out.astype(np.float64)
    assert np.allclose([o64.sum(), (o64 ** 2).sum()], z["moments"], rtol=1e-6)


def test_golden_unet_hook():
    z = load_golden("g8_unet_hook")
    g, d, s = (z[k].astype(np.float32) for k in ("guidance_f16", "blur_f16", "sparse_f16"))
    out = run3(g, d, s, int(z["T"]))
    assert np.abs(out - z["out"]).max() <= 1e-6 and rmse(out, z["out"]) <= RMSE_TOL


PLANS = [
    None,
    dict(force_scalar=1),
    dict(steps_per_launch=1, tile_w=64, tile_h=16, quads_per_thread=1, threads=256),
    dict(steps_per_launch=1, tile_w=32, tile_h=64, quads_per_thread=2, threads=256),
    dict(steps_per_launch=2, tile_w=48, tile_h=24, quads_per_thread=2, threads=256),
    dict(steps_per_launch=3, tile_w=56, tile_h=28, quads_per_thread=2, threads=256),
    dict(steps_per_launch=4, tile_w=56, tile_h=58, quads_per_thread=4, threads=256),
    dict(steps_per_launch=5, tile_w=32, tile_h=54, quads_per_thread=4, threads=256),
    dict(steps_per_launch=6, tile_w=80, tile_h=70, quads_per_thread=4, threads=512),
    dict(steps_per_launch=8, tile_w=48, tile_h=114, quads_per_thread=8, threads=256),
    dict(steps_per_launch=12, tile_w=40, tile_h=84, quads_per_thread=8, threads=256),
    dict(steps_per_launch=24, tile_w=16, tile_h=80, quads_per_thread=8, threads=256),
]


@pytest.mark.parametrize("plan", PLANS, ids=lambda p: "default" if p is None else "-".join(str(v) for v in p.values()))
@pytest.mark.parametrize("sparse", [False, True])
def test_plans_vs_oracle(plan, sparse, c_oracle):
    """Every launch plan (tile shape, quads/thread, fused steps incl. a remainder launch) gives the same answer."""
    B, H, W, T = 3, 100, 148, 7 if plan and plan.get("steps_per_launch", 1) in (2, 3, 4, 5) else 24
    g, d, s = c_oracle.synthetic_inputs(11, B, H, W, 12, 300 if sparse else None)
    want = c_oracle.cspn3_forward(g, d, s, T)
    out = run3(g, d, s, T, plan)
    assert rel_err(out, want) <= REL_TOL and rmse(out, want) <= RMSE_TOL


def test_nan_spread_matches_reference_under_fusion(c_oracle):
    """A 0/0 pixel poisons one more ring per step; temporal blocking must reproduce that exactly."""
    z = load_golden("g2_zero_gates_nan")
    for plan in (None, dict(steps_per_launch=3, tile_w=8, tile_h=8, quads_per_thread=2, threads=256)):
        out = run3(z["guidance"], z["blur"], None, int(z["T"]), plan)
        assert np.array_equal(np.isnan(out), np.isnan(z["out"]))
        assert rel_err(out, z["out"]) <= REL_TOL


def test_full_size_config2_vs_oracle(c_oracle):
    """BASELINE config 2 (B=24, 228x304, 24 steps, 12-channel guidance as the UNet head emits it)."""
    g, d, s = c_oracle.synthetic_inputs(1, 24, 228, 304, 12, 500)
    for sp in (None, s):
        want = c_oracle.cspn3_forward(g, d, sp, 24)
        out = run3(g, d, sp, 24)
        assert rel_err(out, want) <= REL_TOL and rmse(out, want) <= RMSE_TOL


def test_size_independent_properties():
    """At full KITTI size (config 4): constants are fixed points, the map is linear in the depth, and
    obeys the discrete maximum principle (each step is a convex combination of neighbours)."""
    torch.manual_seed(0)
    B, H, W = 2, 352, 1216
    g = torch.randn(B, 12, H, W, device=DEV)
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    with torch.no_grad():
        c = m(g, torch.full((B, 1, H, W), 3.25, device=DEV))
        assert torch.allclose(c, torch.full_like(c, 3.25), rtol=1e-5, atol=0)
        d1 = torch.rand(B, 1, H, W, device=DEV) * 10
        d2 = torch.rand(B, 1, H, W, device=DEV) * 10
        o1, o2, o12 = m(g, d1), m(g, d2), m(g, 0.5 * d1 - 2.0 * d2)
        assert torch.allclose(o12, 0.5 * o1 - 2.0 * o2, rtol=1e-4, atol=1e-4)
        assert o1.min() >= d1.min() - 1e-4 and o1.max() <= d1.max() + 1e-4
        # sparse anchors are restored exactly (CSPN_new.py:90: mask * raw coarse depth)
        sp = d1 * (torch.rand_like(d1) < 0.01)
        os_ = m(g, d1, sp)
        keep = sp > 0
        assert torch.equal(os_[keep], d1[keep])
        # only channels 0..7 matter (unet_cspn_nyu.py:332 emits 12)
        g2 = g.clone()
        g2[:, 8:] = 123.0
        assert torch.equal(m(g2, d1), o1)


@pytest.mark.parametrize("plan", [None, dict(steps_per_launch=1, tile_w=32, tile_h=32, quads_per_thread=1, threads=256),
                                  dict(steps_per_launch=5, tile_w=44, tile_h=30, quads_per_thread=2, threads=512),
                                  dict(steps_per_launch=8, tile_w=40, tile_h=27, quads_per_thread=1, threads=1024)],
                         ids=["default", "s1", "s5nq2", "s8"])
@pytest.mark.parametrize("sparse", [False, True])
def test_from_guidance_equals_prepare_plus_propagate(plan, sparse, c_oracle):
    """The inference entry (weights derived inside every launch) is bit-identical to the two-call form and
    matches the oracle; shapes include image borders inside tiles, a narrow image and a 1-row image."""
    from cspn_monodepth_amd import functional as F
    for (B, H, W, T) in ((3, 100, 148, 24), (2, 37, 8, 7), (1, 1, 12, 3), (1, 5, 4, 6)):
        g, d, s = c_oracle.synthetic_inputs(17, B, H, W, 12, max(2, H * W // 50) if sparse else None)
        gt, dt, st = dev(g), dev(d)[:, 0].contiguous(), (dev(s)[:, 0].contiguous() if sparse else None)
        blend = F.BLEND_SPARSE if sparse else F.BLEND_NONE
        try:
            F.resolve_plan(3, B, H, W, T, False, plan)
        except RuntimeError:
            continue                                   # plan does not fit this tiny shape
        with torch.no_grad():
            w8, _, _ = F.cspn3_prepare(gt)
            a, _ = F.propagate(w8, dt, st, 3, T, blend, plan=plan)
            b, _ = F.propagate_from_guidance(gt, dt, st, T, blend, plan=plan)
            _, hist = F.propagate_from_guidance(gt, dt, st, T, blend, keep_history=True, plan=plan)
        assert torch.equal(a, b) and torch.equal(hist[T - 1], a)
        want = c_oracle.cspn3_forward(g, d, s if sparse else None, T)[:, 0]
        assert rel_err(b.cpu().numpy(), want) <= REL_TOL
        with torch.no_grad():
            c, _ = F.propagate_from_guidance(gt, dt, st, T, blend, plan=plan, publish_weights=False)
        assert torch.equal(a, c)
    # module-level switch (default: from guidance)
    g, d, s = c_oracle.synthetic_inputs(18, 2, 40, 52, 12, 50)
    m = pkg.CSPN_new.AffinityPropagate(9, 3, plan=plan)
    with torch.no_grad():
        ref = m(dev(g), dev(d), dev(s) if sparse else None)
        F.set_from_guidance(False)
        try:
            alt = m(dev(g), dev(d), dev(s) if sparse else None)
        finally:
            F.set_from_guidance(True)
    assert torch.equal(ref, alt)


def test_strided_and_noncontiguous_inputs(c_oracle):
    g, d, s = c_oracle.synthetic_inputs(4, 2, 20, 24, 12, 40)
    want = c_oracle.cspn3_forward(g, d, s, 5)
    gt = dev(np.concatenate([g, g], 1))[:, :12]             # non-contiguous batch stride
    dt = dev(np.concatenate([d, d], 1))[:, :1]
    with torch.no_grad():
        out = pkg.CSPN_new.AffinityPropagate(5, 3)(gt, dt, dev(s)).cpu().numpy()
    assert rel_err(out, want) <= REL_TOL


# ------------------------------------------------------------------------------------------------ K x K
@pytest.mark.parametrize("name", [n for n in golden_names("g6_") if "fp16" not in n])
def test_pac_golden(name):
    z = load_golden(name)
    m = pkg.CSPN_ours.AffinityPropagate(int(z["T"]))
    with torch.no_grad():
        out = m(dev(z["x"]), dev(z["guided"]), sparse_depth=dev(z.get("sparse"))).cpu().numpy()
    assert rel_err(out, z["out"]) <= REL_TOL and rmse(out, z["out"]) <= RMSE_TOL


@pytest.mark.parametrize("K,T,plan", [
    (3, 24, dict(steps_per_launch=4, tile_w=56, tile_h=58, quads_per_thread=4, threads=256)),
    (5, 12, None),
    (5, 12, dict(steps_per_launch=2, tile_w=56, tile_h=40, quads_per_thread=3, threads=256)),
    (5, 12, dict(steps_per_launch=3, tile_w=40, tile_h=36, quads_per_thread=3, threads=256)),
    (7, 6, None),
    (7, 6, dict(steps_per_launch=2, tile_w=32, tile_h=18, quads_per_thread=1, threads=256)),
])
def test_pac_plans_vs_oracle(K, T, plan, c_oracle):
    B, H, W = 2, 60, 88
    gd = c_oracle.hash_normal(K, 1, (B, K * K - 1, H, W))
    x = c_oracle.hash_uniform(K, 2, (B, 1, H, W), 0.0, 10.0)
    s = c_oracle.hash_sparse(K, 3, x, 0.05)
    for sp in (None, s):
        want = c_oracle.pac_forward(x, gd, sp, T)
        with torch.no_grad():
            out = pkg.CSPN_ours.AffinityPropagate(T, plan=plan)(dev(x), dev(gd), sparse_depth=dev(sp)).cpu().numpy()
        assert rel_err(out, want) <= REL_TOL and rmse(out, want) <= RMSE_TOL


def test_pac_fp16_config3():
    """Config 3 (5x5, 12 steps, fp16): reference semantics = fp16 softmax, fp32 state, fp32 output."""
    z = load_golden("g6_k5_t12_fp16")
    x, gd = dev(z["x"]), dev(z["guided"])
    with torch.no_grad():
        out = pkg.CSPN_ours.AffinityPropagate(int(z["T"]))(x, gd)
        out16 = pkg.CSPN_ours.AffinityPropagate(int(z["T"]), state_dtype=None)(x, gd)
    assert out.dtype == torch.float32 and out16.dtype == torch.float16
    # fp16 softmax weights carry ~5e-4 relative rounding; the reference rounds them the same way
    assert rel_err(out.cpu().numpy(), z["out"]) <= 2e-3 and rmse(out.cpu().numpy(), z["out"]) <= 2e-3
    assert rmse(out16.float().cpu().numpy(), z["out"]) <= 1e-2


# ------------------------------------------------------------------------------------------------ metrics
def test_metrics_kernel_golden():
    z = load_golden("g7_metrics")
    ev = pkg.evaluation
    sums = ev.metric_sums(dev(z["pred"]), dev(z["target"]))
    want = orc.metric_sums(z["pred"], z["target"])
    assert np.allclose(sums.cpu().numpy(), want, rtol=1e-5)
    fin = ev.finalize_metrics(sums.cpu())
    for k, v in zip(ev.METRIC_NAMES, z["metrics"]):
        assert np.isclose(fin[k], v, rtol=1e-5), k
    assert fin["count"] == int((z["target"] > 0).sum())
    # accumulating form: two batches into one [nslots, 10] accumulator, no host sync in between
    acc = ev.new_accumulator(DEV)
    ev.metric_sums(dev(z["pred"]), dev(z["target"]), out=acc)
    ev.metric_sums(dev(z["pred"]), dev(z["target"]), out=acc)
    assert np.allclose(acc.sum(0).cpu().numpy(), 2 * want, rtol=1e-5)
    assert ev.finalize_metrics(acc.cpu())["count"] == 2 * fin["count"]
    # a full-size batch (odd length -> scalar tail) against the numpy oracle
    rng = np.random.default_rng(0)
    t = rng.uniform(0.5, 10, 24 * 228 * 304 + 3).astype(np.float32)
    t[rng.random(t.size) < 0.05] = 0
    p = np.abs(t + rng.normal(0, 0.1, t.size).astype(np.float32)) + 0.01
    got = ev.metric_sums(dev(p.astype(np.float32)), dev(t)).cpu().numpy()
    assert np.allclose(got, orc.metric_sums(p.astype(np.float32), t), rtol=2e-5)


# ------------------------------------------------------------------------------------------------ fp16 storage
def _half_close(got, want, tol=4e-3):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.abs(got - want).max()) <= tol * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("shape", [(2, 40, 52), (1, 7, 18), (1, 5, 9), (2, 33, 64)])
@pytest.mark.parametrize("plan", [None, dict(steps_per_launch=1, tile_w=32, tile_h=32, quads_per_thread=1, threads=256),
                                  dict(steps_per_launch=4, tile_w=24, tile_h=26, quads_per_thread=2, threads=256)],
                         ids=["default", "s1", "s4nq2"])
def test_cspn3_fp16_storage(shape, plan, c_oracle):
    """fp16 tap volume (pair-interleaved layout, incl. the padded HW % 4 != 0 case and the scalar kernel for
    W % 4 != 0) with fp16 depth state: against the fp32 oracle on the fp16-rounded inputs."""
    B, H, W = shape
    g, d, s = c_oracle.synthetic_inputs(23, B, H, W, 12, max(2, H * W // 40))
    g16, d16, s16 = (a.astype(np.float16) for a in (g, d, s))
    want = c_oracle.cspn3_forward(g16.astype(np.float32), d16.astype(np.float32), s16.astype(np.float32), 8)
    with torch.no_grad():
        out = pkg.CSPN_new.AffinityPropagate(8, 3, plan=plan)(dev(g16), dev(d16), dev(s16))
    assert out.dtype == torch.float16
    assert _half_close(out.float().cpu().numpy(), want)


@pytest.mark.parametrize("K,T", [(3, 10), (5, 12), (7, 4)])
@pytest.mark.parametrize("state", ["reference", None])
def test_pac_fp16_storage(K, T, state, c_oracle):
    B, H, W = 2, 36, 44
    gd = c_oracle.hash_normal(K, 1, (B, K * K - 1, H, W)).astype(np.float16)
    x = c_oracle.hash_uniform(K, 2, (B, 1, H, W), 0.0, 10.0).astype(np.float16)
    s = c_oracle.hash_sparse(K, 3, x.astype(np.float32), 0.05).astype(np.float16)
    want = c_oracle.pac_forward(x.astype(np.float32), gd.astype(np.float32), s.astype(np.float32), T)
    plans = [None, dict(steps_per_launch=2, tile_w=24, tile_h=14 if K == 7 else 20, quads_per_thread=1, threads=256)]
    for plan in plans:
        with torch.no_grad():
            out = pkg.CSPN_ours.AffinityPropagate(T, plan=plan, state_dtype=state)(dev(x), dev(gd), sparse_depth=dev(s))
        assert out.dtype == (torch.float32 if state == "reference" else torch.float16)
        assert _half_close(out.float().cpu().numpy(), want, 6e-3)


@pytest.mark.parametrize("sparse", [False, True])
def test_scored_forward_equals_forward_plus_metrics(sparse, c_oracle):
    """forward_scored (metrics fused into the last propagation launch) = forward + evaluation.metric_sums."""
    ev = pkg.evaluation
    for (B, H, W, T, plan) in ((3, 100, 148, 24, None), (2, 37, 8, 7, None), (1, 9, 18, 5, None),   # last: W%4 != 0
                               (2, 60, 64, 9, dict(steps_per_launch=3, tile_w=32, tile_h=28, quads_per_thread=2, threads=256))):
        g, d, s = c_oracle.synthetic_inputs(29, B, H, W, 12, max(2, H * W // 50))
        tgt = np.maximum(d + 0.1 * c_oracle.hash_normal(30, 9, d.shape), 0.0).astype(np.float32)
        m = pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)
        sp = dev(s) if sparse else None
        with torch.no_grad():
            ref = m(dev(g), dev(d), sp)
            want = ev.metric_sums(ref, dev(tgt))
            acc = ev.new_accumulator(DEV)
            out = m.forward_scored(dev(g), dev(d), sp, dev(tgt), acc)
        assert torch.equal(out, ref)
        assert np.allclose(acc.sum(0).cpu().numpy(), want.cpu().numpy(), rtol=1e-5)
    # K x K, fp32 and fp16 storage
    K, T, B, H, W = 5, 6, 2, 40, 48
    gd = c_oracle.hash_normal(31, 1, (B, 24, H, W)); x = c_oracle.hash_uniform(31, 2, (B, 1, H, W), 0.0, 10.0)
    tgt = np.maximum(x + 0.1 * c_oracle.hash_normal(32, 9, x.shape), 0.0).astype(np.float32)
    for dt in (np.float32, np.float16):
        m = pkg.CSPN_ours.AffinityPropagate(T, state_dtype=None)
        with torch.no_grad():
            ref = m(dev(x.astype(dt)), dev(gd.astype(dt)))
            want = ev.metric_sums(ref, dev(tgt.astype(dt)))
            acc = ev.new_accumulator(DEV)
            out = m.forward_scored(dev(x.astype(dt)), dev(gd.astype(dt)), None, dev(tgt.astype(dt)), acc)
        assert torch.equal(out, ref)
        assert np.allclose(acc.sum(0).cpu().numpy(), want.cpu().numpy(), rtol=1e-4 if dt == np.float16 else 1e-5)


# ------------------------------------------------------------------------------------------------ runtime behaviour
def test_thread_safety_and_streams(c_oracle):
    """Re-entrancy: concurrent calls from several host threads, each on its own HIP stream
    (the reference's DataParallel pattern, network/libs/base/encoding.py:160-186), give the serial results."""
    import threading
    B, H, W, T = 2, 60, 76, 12
    cases = []
    for i in range(4):
        g, d, s = c_oracle.synthetic_inputs(60 + i, B, H, W, 12, 60)
        cases.append((dev(g), dev(d), dev(s)))
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    with torch.no_grad():
        serial = [m(*c).clone() for c in cases]
    torch.cuda.synchronize()
    results, errors = [None] * len(cases), []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(20):
                    out = m(*cases[i])
                st.synchronize()
            results[i] = out
        except Exception as e:       # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(cases))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors
    for a, b in zip(serial, results):
        assert torch.equal(a, b)


def test_hip_graph_capture_and_replay(c_oracle):
    """The engine only enqueues kernels on the caller's stream (no allocation, no sync inside the .so), so a whole
    forward can be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed."""
    B, H, W, T = 3, 228, 304, 24
    g, d, s = c_oracle.synthetic_inputs(70, B, H, W, 12, 500)
    gt, dt, st = dev(g), dev(d), dev(s)
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    with torch.no_grad():
        ref = m(gt, dt, st).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                m(gt, dt, st)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = m(gt, dt, st)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        dt.mul_(0.5)                       # new input in the captured buffers
        graph.replay()
        torch.cuda.synchronize()
        assert torch.allclose(out, 0.5 * ref, rtol=1e-5, atol=1e-6)


def test_prepare_normalisation_tracks_ieee_division():
    """cspn3_prepare's shared-reciprocal normalisation (q = a * refined 1/S): S itself is exact in the reference's
    summation order, every weight is within 2 ulp of the IEEE quotient, and zero / inf / nan / extreme-exponent
    operands produce exactly the IEEE special values (they take the true division or the NaN-through-rcp route)."""
    from cspn_monodepth_amd import functional as F
    torch.manual_seed(3)
    B, H, W = 2, 64, 96
    g = torch.randn(B, 8, H, W, device=DEV)
    specials = torch.tensor([0.0, 1e-30, 1e30, 1e-41, 3e38, 1e-20, 1e20, float("inf"), float("nan"), 2.0 ** -60, 2.0 ** 60],
                            device=DEV)
    idx = torch.randint(0, g.numel(), (4000,), device=DEV)
    g.view(-1)[idx] = specials[torch.randint(0, len(specials), (4000,), device=DEV)]
    g[0, :, 10:14, 10:14] = 0.0                                   # all-zero neighbourhoods -> 0/0
    g[1, :, 30:34, 40:48] *= 1e-25                                # uniformly tiny neighbourhoods
    w8, S, _ = F.cspn3_prepare(g, want_s=True)
    offs = [(+1, +1), (+1, 0), (+1, -1), (0, +1), (0, -1), (-1, +1), (-1, 0), (-1, -1)]     # o_k of reference plane k
    A = []
    for k, (dy, dx) in enumerate(offs):
        a = torch.zeros(B, H, W, device=DEV)
        ys, yd = (slice(dy, H), slice(0, H - dy)) if dy > 0 else ((slice(0, H + dy), slice(-dy, H)) if dy < 0 else (slice(0, H), slice(0, H)))
        xs, xd = (slice(dx, W), slice(0, W - dx)) if dx > 0 else ((slice(0, W + dx), slice(-dx, W)) if dx < 0 else (slice(0, W), slice(0, W)))
        a[:, yd, xd] = g[:, k, ys, xs].abs()
        A.append(a)
    Sref = A[0].clone()
    for k in range(1, 8):
        Sref = Sref + A[k]
    assert torch.equal(torch.nan_to_num(S, nan=-1.0), torch.nan_to_num(Sref, nan=-1.0))
    for j in range(8):
        ref = A[7 - j] / Sref                                      # tap j <-> reference channel 7-j
        got = w8[:, j]
        assert torch.equal(torch.isnan(got), torch.isnan(ref)), j
        gi = torch.nan_to_num(got, nan=0.0, posinf=9e9, neginf=-9e9).view(torch.int32)      # quotients are >= 0:
        ri = torch.nan_to_num(ref, nan=0.0, posinf=9e9, neginf=-9e9).view(torch.int32)      # bit patterns are ordered
        assert int((gi - ri).abs().max()) <= 2, (j, int((gi - ri).abs().max()))
        special = ~torch.isfinite(Sref) | (Sref == 0) | (Sref < 2.0 ** -100) | (Sref > 2.0 ** 100)
        assert torch.equal(gi[special], ri[special]), j


def test_graphed_forward_helper(c_oracle):
    g, d, s = c_oracle.synthetic_inputs(71, 1, 64, 96, 12, 80)
    tgt = np.maximum(d + 0.1 * c_oracle.hash_normal(72, 9, d.shape), 0.0).astype(np.float32)
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    acc = pkg.evaluation.new_accumulator(DEV)
    gt, dt, st, tt = dev(g), dev(d), dev(s), dev(tgt)
    with torch.no_grad():
        ref = m(gt, dt, st).clone()
        want = pkg.evaluation.metric_sums(ref, tt)
    graphed = pkg.graphs.GraphedForward(lambda a, b, c, t: m.forward_scored(a, b, c, t, acc), gt, dt, st, tt)
    acc.zero_()
    out = graphed(gt, dt, st, tt)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert np.allclose(acc.sum(0).cpu().numpy(), want.cpu().numpy(), rtol=1e-5)
    out2 = graphed(gt, dt * 0.5, st * 0.5, tt)         # fresh inputs are copied into the captured buffers
    torch.cuda.synchronize()
    assert torch.allclose(out2, 0.5 * ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("W", [17, 65, 207, 3, 1])
def test_any_width_row_padding_is_exact(W, c_oracle):
    """Widths with W % 4 != 0 run on the quad kernels through row padding (W_valid): bit-identical to the
    generic one-pixel-per-thread kernels for the 3x3 variant, oracle parity for both variants, NaN semantics
    of degenerate shapes preserved, gradients correct through the padding."""
    B, H, T = 2, 23, 9
    g, d, s = c_oracle.synthetic_inputs(81, B, H, W, 12, max(2, H * W // 20))
    vec, gen = pkg.CSPN_new.AffinityPropagate(T, 3), pkg.CSPN_new.AffinityPropagate(T, 3, plan=dict(force_scalar=1))
    for sp in (None, s):
        want = c_oracle.cspn3_forward(g, d, sp, T)
        with torch.no_grad():
            a, b = vec(dev(g), dev(d), dev(sp)), gen(dev(g), dev(d), dev(sp))
        assert a.shape == (B, 1, H, W) and a.is_contiguous() is not None
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
        assert rel_err(a.cpu().numpy(), want) <= REL_TOL
    if W >= 3:
        gt, dt = dev(g).requires_grad_(True), dev(d).requires_grad_(True)
        cot = c_oracle.hash_normal(82, 9, (B, 1, H, W))
        vec(gt, dt, dev(s)).backward(dev(cot))
        wg, wd = c_oracle.cspn3_backward(g, d, s, cot, T, np.float64)
        assert gt.grad.shape == gt.shape and dt.grad.shape == dt.shape
        assert np.abs(gt.grad.cpu().numpy() - wg).max() <= 5e-4 * max(1.0, np.abs(wg).max())
        assert np.abs(dt.grad.cpu().numpy() - wd).max() <= 5e-5 * max(1.0, np.abs(wd).max())
        # scored forward on an odd width
        tgt = np.maximum(d + 0.1 * c_oracle.hash_normal(84, 9, d.shape), 0.0).astype(np.float32)
        acc = pkg.evaluation.new_accumulator(DEV)
        with torch.no_grad():
            ref = vec(dev(g), dev(d), dev(s))
            out = vec.forward_scored(dev(g), dev(d), dev(s), dev(tgt), acc)
            want_m = pkg.evaluation.metric_sums(ref, dev(tgt))
        assert torch.equal(out, ref) and np.allclose(acc.sum(0).cpu().numpy(), want_m.cpu().numpy(), rtol=1e-5)
    # K x K
    gd = c_oracle.hash_normal(83, 1, (B, 24, H, W)); x = c_oracle.hash_uniform(83, 2, (B, 1, H, W), 0.0, 10.0)
    for sp in (None, c_oracle.hash_sparse(83, 3, x, 0.05)):
        want = c_oracle.pac_forward(x, gd, sp, 6)
        with torch.no_grad():
            out = pkg.CSPN_ours.AffinityPropagate(6)(dev(x), dev(gd), sparse_depth=dev(sp))
        assert out.shape == (B, 1, H, W) and rel_err(out.cpu().numpy(), want) <= REL_TOL
    if W >= 3:
        xt, gdt = dev(x).requires_grad_(True), dev(gd).requires_grad_(True)
        cot = c_oracle.hash_normal(85, 9, (B, 1, H, W))
        pkg.CSPN_ours.AffinityPropagate(4)(xt, gdt).backward(dev(cot))
        wx, wgd = orc.pac_backward(x, gd, None, cot, 4, np.float64)
        assert np.abs(xt.grad.cpu().numpy() - wx).max() <= 5e-5 * max(1.0, np.abs(wx).max())
        assert np.abs(gdt.grad.cpu().numpy() - wgd).max() <= 5e-4 * max(1.0, np.abs(wgd).max())


def test_api_edge_cases(c_oracle):
    """prop_time = 0 (identity), non-default device index handling, dtype / shape validation errors."""
    g, d, s = c_oracle.synthetic_inputs(91, 2, 12, 16, 12, 20)
    with torch.no_grad():
        out = pkg.CSPN_new.AffinityPropagate(0, 3)(dev(g), dev(d), dev(s))
    assert torch.equal(out, dev(d))                               # the reference returns blur_depth unchanged
    gt, dt = dev(g).requires_grad_(True), dev(d).requires_grad_(True)
    out = pkg.CSPN_new.AffinityPropagate(1, 3)(gt, dt)
    out.sum().backward()
    assert torch.isfinite(gt.grad).all() and torch.isfinite(dt.grad).all()
    m = pkg.CSPN_new.AffinityPropagate(3, 3)
    with pytest.raises(TypeError):
        m(dev(g).half(), dev(d))                                  # mixed dtypes
    with pytest.raises(TypeError):
        m(dev(g).to(torch.bfloat16), dev(d).to(torch.bfloat16))   # unsupported dtype
    with pytest.raises(ValueError):
        m(dev(g)[:, :7], dev(d))                                  # fewer than 8 guidance channels
    with pytest.raises(ValueError):
        m(dev(g), dev(d)[:, :, :-1])                              # shape mismatch
    with pytest.raises(ValueError):
        pkg.CSPN_ours.AffinityPropagate(3)(dev(d), dev(g)[:, :10])   # 10 channels is not K*K-1
    sp = dev(s).requires_grad_(True)                              # sparse depth gets no gradient (sign())
    pkg.CSPN_new.AffinityPropagate(2, 3)(gt, dt, sp).sum().backward()
    assert sp.grad is None
