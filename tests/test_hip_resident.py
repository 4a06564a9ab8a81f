"""GPU tests of the weight-resident single-launch forward (cspn3_forward_resident, csrc/cspn_resident.hip): one launch
for all T steps, weights in registers, tile borders exchanged between co-resident workgroups.  It must give the SAME
BITS as the multi-launch schedule (same weight arithmetic, same FMA order) and match the oracle / the reference goldens.

Reference: network/libs/post_process/CSPN_new.py:26-92."""
import numpy as np
import pytest
import torch

import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
from conftest import bits_equal, lds_poison, golden_names, load_golden, rel_err, rmse

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


class resident(object):
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = F._RESIDENT_MODE
        F.set_resident(self.mode)

    def __exit__(self, *exc):
        F.set_resident(self.prev)
        return False


def both(g, d, s, T):
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    with torch.no_grad():
        with resident("off"):
            ref = m(dev(g), dev(d), dev(s))
        with resident("on"):
            assert F.resident_supported(dev(g), dev(d)[:, 0], None if s is None else dev(s)[:, 0], T) is not None
            out = m(dev(g), dev(d), dev(s))
    torch.cuda.synchronize()
    F.check_resident_errors()
    return out, ref


SHAPES = [(24, 228, 304, 24), (3, 228, 304, 24), (1, 352, 1216, 24), (8, 352, 1216, 24), (1, 228, 304, 24), (2, 13, 20, 24),
          (5, 60, 64, 7), (2, 37, 8, 1), (1, 1, 12, 3), (3, 100, 148, 9), (1, 5, 4, 6), (30, 120, 160, 17),
          (97, 228, 304, 24),           # five resident launches per call (the auto policy has no cap on their number)
          # single-phase launches (T <= 8) whose halo is deeper than a tile: the region must reach y0 - hyw, not stop at the
          # previous tile's start (round 3: found by the K x K fuzz, the same origin rule lived here)
          (1, 60, 64, 7), (1, 59, 40, 7), (2, 23, 104, 7), (1, 45, 96, 6), (4, 72, 112, 8)]


@pytest.mark.parametrize("B,H,W,T", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_resident_equals_multi_launch_bit_for_bit(B, H, W, T, sparse, c_oracle):
    g, d, s = c_oracle.synthetic_inputs(70 + B + T, B, H, W, 12, max(2, H * W // 140) if sparse else None)
    out, ref = both(g, d, s, T)
    assert bits_equal(out, ref, T=T, sparse=sparse)
    if B * H * W <= 24 * 228 * 304:
        want = c_oracle.cspn3_forward(g, d, s, T)
        assert rel_err(out.cpu().numpy(), want) <= 1e-5 and rmse(out.cpu().numpy(), want) <= 1e-4


# Shapes whose resident tiling leaves the last strip of a thread column partly PAST the region (wr % NQ != 0): those rows are
# computed with zero taps and serve as the lower neighbour of the region's last row.  KITTI B = 8 is the production one.
# (B, H, W, T): NQ, wr on 256 CUs — 8x352x1216: 5, 58; 4x228x304: 2, 51; 12x228x304: 3, 47; 16x228x304: 4, 79; 20x240x320: 5, 54;
# 6x256x512x12: 4, 51; 2x300x400: 2, 47; then the shapes of configs 2 / 4 / 5 (wr % NQ == 0) and a single-phase one.
STALE_LDS_SHAPES = [(8, 352, 1216, 24), (4, 228, 304, 24), (12, 228, 304, 24), (16, 228, 304, 24), (20, 240, 320, 24), (6, 256, 512, 12),
                    (2, 300, 400, 24), (24, 228, 304, 24), (3, 228, 304, 24), (1, 352, 1216, 24), (5, 60, 64, 7)]


@pytest.mark.parametrize("B,H,W,T", STALE_LDS_SHAPES, ids=["x".join(map(str, s)) for s in STALE_LDS_SHAPES])
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_resident_result_does_not_depend_on_stale_lds(B, H, W, T, sparse, c_oracle):
    """Round 5, the root cause of round 4's one-off mismatch of [sparse-8x352x1216x24]: LDS is not cleared between kernels, and
    the blended CLEAN instances of cspn3_resident added a never-written private slot (m * d0) to the rows past the region; for a
    bottom-edge tile that sum is the zero padding below the image, and a NaN / Inf bit pattern left there by an earlier kernel
    (0 x NaN) spread into the last 24 rows of every image.  Only shapes with wr % NQ != 0 were exposed — KITTI B = 8 (config 4 on
    one GPU) is the one among the tested shapes.  With a NaN fill of the whole LDS in front of EVERY launch (resident and
    multi-launch alike) the results must still be the bits of the unpoisoned multi-launch schedule."""
    g, d, s = c_oracle.synthetic_inputs(170 + B + T, B, H, W, 12, max(2, H * W // 140) if sparse else None)
    _, ref = both(g, d, s, T)
    with lds_poison():
        out, ref_p = both(g, d, s, T)
    assert bits_equal(ref_p, ref, T=T, sparse=sparse, which="multi-launch under poison")
    assert bits_equal(out, ref, T=T, sparse=sparse, which="resident under poison")


@pytest.mark.parametrize("name", golden_names("g1_") + golden_names("g2_") + ["g8_unet_hook"])
def test_resident_on_reference_goldens(name):
    """Small / degenerate / NaN-spreading / negative-sparse cases captured from the reference, through the resident launch."""
    z = load_golden(name)
    if name == "g8_unet_hook":
        g, d, s = (z[k].astype(np.float32) for k in ("guidance_f16", "blur_f16", "sparse_f16"))
    else:
        g, d, s = z["guidance"], z["blur"], z.get("sparse")
    T = int(z["T"])
    if g.shape[-1] % 4:                                   # odd widths: the module pads the rows and passes W_valid
        m = pkg.CSPN_new.AffinityPropagate(T, 3)
        with torch.no_grad(), resident("on"):
            out = m(dev(g), dev(d), dev(s)).cpu().numpy()
    else:
        out = both(g, d, s, T)[0].cpu().numpy()
    if name == "g8_unet_hook":
        assert np.abs(out - z["out"]).max() <= 1e-6
    else:
        assert rel_err(out, z["out"]) <= 1e-5, name


def test_resident_scored_and_phase_lengths(c_oracle):
    """forward_scored through the resident launch = forward + metric sums; every steps-per-phase gives the same bits."""
    ev = pkg.evaluation
    B, H, W, T = 24, 228, 304, 24
    g, d, s = c_oracle.synthetic_inputs(91, B, H, W, 12, 500)
    tgt = np.maximum(d + 0.1 * c_oracle.hash_normal(92, 9, d.shape), 0.0).astype(np.float32)
    tgt[c_oracle.hash_uniform(93, 9, d.shape) < 0.05] = 0.0
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    gt, dt, tt = dev(g), dev(d), dev(tgt)
    for sp in (None, dev(s)):
        with torch.no_grad():
            with resident("off"):
                acc0 = ev.new_accumulator(DEV)
                ref = m.forward_scored(gt, dt, sp, tt, acc0)
            with resident("on"):
                acc1 = ev.new_accumulator(DEV)
                out = m.forward_scored(gt, dt, sp, tt, acc1)
        assert bits_equal(out, ref)
        assert np.allclose(acc1.sum(0).cpu().numpy(), acc0.sum(0).cpu().numpy(), rtol=1e-6)
        for S in (4, 6, 8):
            with torch.no_grad():
                o = F.forward_resident(gt, dt[:, 0], None if sp is None else sp[:, 0], T, int(sp is not None), steps_per_phase=S)
            assert bits_equal(o, ref[:, 0]), S
    F.check_resident_errors()


def test_resident_launches_from_several_streams_are_serialised(c_oracle):
    """Two host threads on two HIP streams: the module orders the resident launches after one another (they must
    never overlap on a device); results equal the serial ones."""
    import threading
    B, H, W, T = 24, 228, 304, 24
    cases = []
    for i in range(2):
        g, d, s = c_oracle.synthetic_inputs(95 + i, B, H, W, 12, 300)
        cases.append((dev(g), dev(d), dev(s)))
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    with torch.no_grad(), resident("on"):
        serial = [m(*c) for c in cases]
        res = [None] * len(cases)

        def work(i):
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(20):
                    res[i] = m(*cases[i])
            st.synchronize()

        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
        [t.start() for t in ths]
        [t.join() for t in ths]
    torch.cuda.synchronize()
    F.check_resident_errors()
    for a, b_ in zip(res, serial):
        assert bits_equal(a, b_)


def test_resident_timeout_is_repaired_not_a_hang(c_oracle):
    """A neighbour wait that gives up (spin limit forced to 1 poll) must end the launch; the next launch on the device finds the
    error word and re-runs the failed call on the multi-launch schedule INTO THE SAME TENSOR — no raise, the same bits.  (The HOST
    repair: the call is made without the device-side guard, as scored calls are.)"""
    B, H, W, T = 24, 228, 304, 24
    g, d, _ = c_oracle.synthetic_inputs(99, B, H, W, 12, None)
    gt, dt = dev(g), dev(d)[:, 0].contiguous()
    F.ensure_resident_ok()                                  # an empty journal: the launch below is not the one a full journal waits for
    with torch.no_grad():
        with resident("off"):
            ref = pkg.CSPN_new.AffinityPropagate(T, 3)(gt, dev(d))
        broken = F.forward_resident(gt, dt, None, T, 0, spin_limit=1, guard=0)
        torch.cuda.synchronize()
        if F.resident_fallbacks() == 0:                                 # nobody has looked yet: the tiles that gave up read as NaN,
            assert bool(torch.isnan(broken).any()) and F._holds_poison(broken)      # with the payload the host recognises
        # (the host may have looked already: recording the call's completion mark can take longer than the 20 us the launch needs
        #  to give up, and the journal then repairs at once)
        out = F.forward_resident(gt, dt, None, T, 0)                   # finds the error word, repairs `broken`, then runs
        torch.cuda.synchronize()
    assert F.resident_fallbacks() == 1
    assert bits_equal(broken, ref[:, 0]) and bits_equal(out, ref[:, 0])
    F.ensure_resident_ok()


def test_resident_halo_comes_from_adjacent_tiles_only(c_oracle):
    """Regression (round 2): a last tile row / column cut short by the image edge must not shift its region further than
    the previous tile's start — the halo would come from a tile two away, which the exchange does not wait for.
    KITTI B=1 (11x20 tiles of 112x18, last row 10 pixels high) and a few ragged shapes, against the oracle."""
    for (B, H, W) in ((1, 352, 1216), (2, 100, 148), (1, 75, 300), (3, 41, 52)):
        for T in (24, 9):
            g, d, s = c_oracle.synthetic_inputs(120 + H, B, H, W, 12, 200)
            want = c_oracle.cspn3_forward(g, d, s, T)
            out = both(g, d, s, T)[0].cpu().numpy()
            assert rel_err(out, want) <= 1e-5, (B, H, W, T)


def test_resident_does_not_depend_on_leftover_state(c_oracle):
    """Regression (round 2): LDS and the exchange planes keep whatever the previous launch left there.  Poison both with a
    NaN-producing call, then alternate between different inputs: every output must still equal the multi-launch result
    (an uninitialised ring row of the second LDS buffer once leaked the previous kernel's values into edge tiles)."""
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    for (B, H, W) in ((24, 228, 304), (8, 352, 1216), (3, 100, 148)):
        sets = []
        for k in range(3):
            g, d, s = c_oracle.synthetic_inputs(200 + k + B, B, H, W, 12, 300)
            sets.append((dev(g), dev(d), dev(s) if k == 1 else None))
        poison = (torch.zeros_like(sets[0][0]), torch.full_like(sets[0][1], float("nan")), None)   # 0/0 weights, NaN depth
        with torch.no_grad():
            with resident("off"):
                refs = [m(*c) for c in sets]
            with resident("on"):
                for rep in range(4):
                    m(*poison)
                    for k in (2, 0, 1, 1, 2, 0):
                        assert bits_equal(m(*sets[k]), refs[k]), (B, H, W, rep, k)
    F.check_resident_errors()


@pytest.mark.parametrize("B,H,W", [(3, 100, 148), (24, 228, 304), (8, 352, 1216)], ids=lambda v: str(v))
def test_resident_launch_replays_from_a_hip_graph(B, H, W, c_oracle):
    """A replay cannot bring a fresh flag sequence number: the capture records a memset of the workspace's control words in
    front of the launch and a constant sequence number (functional._resident_launch).  Replays with NEW input values must
    equal the eager results bit for bit — also when eager resident launches (their own, growing sequence numbers) run in
    between, and for the two-launch plan of the KITTI batch; the scored variant accumulates the same sums."""
    T = 24
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    ins = [c_oracle.synthetic_inputs(130 + k, B, H, W, 12, 300) for k in range(3)]
    gt, dt, st = (dev(x).clone() for x in ins[0])
    tgt = dev(np.maximum(ins[0][1] + 0.05, 0.0).astype(np.float32))
    acc = pkg.evaluation.new_accumulator(DEV)
    with torch.no_grad(), resident("on"):
        assert F.resident_supported(gt, dt[:, 0], st[:, 0], T, None, tgt[:, 0]) is not None
        graphed = pkg.graphs.GraphedForward(lambda: m.forward_scored(gt, dt, st, tgt, acc))
        for k in (1, 2, 0, 1):
            g, d, s = ins[k]
            gt.copy_(dev(g)); dt.copy_(dev(d)); st.copy_(dev(s))
            acc.zero_()
            out = graphed(copy_inputs=False).clone()
            got = acc.sum(0).cpu().numpy()
            acc0 = pkg.evaluation.new_accumulator(DEV)
            ref = m.forward_scored(gt, dt, st, tgt, acc0)                      # eager resident launch in between
            assert bits_equal(out, ref), k
            assert np.allclose(got, acc0.sum(0).cpu().numpy(), rtol=1e-6)
        torch.cuda.synchronize()
    want = c_oracle.cspn3_forward(*ins[1], T)
    assert rel_err(out.cpu().numpy(), want) <= 1e-5
    F.check_resident_errors()


@pytest.mark.parametrize("B,H,W,T", [(24, 228, 304, 24), (3, 228, 304, 24), (1, 352, 1216, 24), (2, 37, 8, 7), (5, 60, 64, 9)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_resident_training_forward_publishes_what_the_backward_needs(B, H, W, T, sparse, c_oracle):
    """The training form of the resident launch (history planes, weights and S published once) against the multi-launch
    training forward: every history plane, the tap volume and S bit for bit; gradients through the module against the
    fp64 oracle."""
    g, d, s = c_oracle.synthetic_inputs(150 + B + T, B, H, W, 12, max(2, H * W // 140) if sparse else None)
    gt, dt = dev(g), dev(d)[:, 0].contiguous()
    st = dev(s)[:, 0].contiguous() if sparse else None
    blend = F.BLEND_SPARSE if sparse else F.BLEND_NONE
    with torch.no_grad():
        _, hist0, w0, S0 = F.propagate_from_guidance(gt, dt, st, T, blend, keep_history=True, return_weights=True)
        out1, hist1, w1, S1 = F.forward_resident(gt, dt, st, T, int(sparse), keep_history=True)
    assert bits_equal(hist1, hist0) and bits_equal(w1, w0) and bits_equal(S1, S0) and bits_equal(out1, hist0[T - 1])
    # through autograd (mode "on": the training forward takes the resident launch)
    cot = c_oracle.hash_normal(151, 9, (B, 1, H, W))
    wg, wd = c_oracle.cspn3_backward(g, d, s, cot, T, np.float64)
    ga, da = dev(g).requires_grad_(True), dev(d).requires_grad_(True)
    with resident("on"):
        assert F.resident_supported(ga, da[:, 0], st, T) is not None
        out = pkg.CSPN_new.AffinityPropagate(T, 3)(ga, da, dev(s))
    out.backward(dev(cot))
    close = lambda a, b, tol: float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max()))   # noqa: E731
    assert close(ga.grad.cpu().numpy(), wg, 5e-4) and close(da.grad.cpu().numpy(), wd, 5e-5)
    F.check_resident_errors()


@pytest.mark.parametrize("B,H,W,T", [(24, 228, 304, 24), (3, 228, 304, 24), (1, 352, 1216, 24), (2, 37, 8, 7), (5, 60, 64, 9)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_resident_reverse_sweep_equals_multi_launch(B, H, W, T, sparse, c_oracle):
    """cspn3_transposed_resident (the backward's reverse sweep with the transposed taps resident in registers) against
    cspn_propagate_transposed on the same tap volume and cotangent: every G_t plane bit for bit."""
    g, d, s = c_oracle.synthetic_inputs(170 + B + T, B, H, W, 12, max(2, H * W // 140) if sparse else None)
    cot = dev(c_oracle.hash_normal(171, 9, (B, H, W)))
    with torch.no_grad():
        w8, _, _ = F.cspn3_prepare(dev(g))
    sp = dev(s)[:, 0].contiguous() if sparse else None
    with resident("off"):
        _, ref = F._reverse_sweep(w8, 3, T, sp, cot, None)
    with resident("on"):
        _, out = F._reverse_sweep(w8, 3, T, sp, cot, None)
        direct = F.transposed_resident(w8, cot, sp, T)
    assert bits_equal(out, ref) and bits_equal(direct, ref)
    F.check_resident_errors()


@pytest.mark.parametrize("B,H,W,T", [(24, 228, 304, 24), (3, 228, 304, 24), (1, 352, 1216, 24), (2, 37, 8, 7), (5, 60, 64, 9), (2, 7, 12, 3)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_volume_free_reverse_sweep_equals_the_sweep_on_the_published_volume(B, H, W, T, sparse, c_oracle):
    """cspn3_transposed_resident_guidance rebuilds the transposed taps |g_j[p]| / S[p + off_j] from the raw guidance and the
    normaliser S (ABI 9) instead of gathering them from the 8-plane volume: every G_t plane bit for bit equal to the sweep on the
    volume — also with a 12-channel guidance read through its strides, and with the S the TRAINING forward publishes when it is
    told not to write the volume (publish_weights=False), whose history and S equal the publishing launch's."""
    g, d, s = c_oracle.synthetic_inputs(190 + B + T, B, H, W, 12, max(2, H * W // 140) if sparse else None)
    cot = dev(c_oracle.hash_normal(191, 9, (B, H, W)))
    gd, d0 = dev(g), dev(d)[:, 0].contiguous()
    sp = dev(s)[:, 0].contiguous() if sparse else None
    with torch.no_grad():
        w8, S, _ = F.cspn3_prepare(gd, want_s=True)
    with resident("off"):
        _, ref = F._reverse_sweep(w8, 3, T, sp, cot, None)
    with resident("on"):
        direct = F.transposed_resident_guidance(gd, S, cot, sp, T)
        _, via = F._reverse_sweep(None, 3, T, sp, cot, None, guidance_S=(gd, S))
        assert bits_equal(direct, ref) and bits_equal(via, ref)
        if F.resident_supported(gd, d0, sp, T) is not None:
            blend = F.BLEND_SPARSE if sparse else F.BLEND_NONE
            o1, h1, w1, S1 = F.forward_resident(gd, d0, sp, T, blend, keep_history=True)
            o2, h2, w2, S2 = F.forward_resident(gd, d0, sp, T, blend, keep_history=True, publish_weights=False)
            assert w2 is None and bits_equal(h1, h2) and bits_equal(S1, S2) and bits_equal(w1, w8) and bits_equal(S1, S)
            assert bits_equal(F.transposed_resident_guidance(gd, S2, cot, sp, T), ref)
    with resident("off"):          # no resident sweep: the volume is rebuilt by cspn3_prepare, the streaming launches run on it
        _, off = F._reverse_sweep(None, 3, T, sp, cot, None, guidance_S=(gd, S))
    assert bits_equal(off, ref)
    F.check_resident_errors()


@pytest.mark.parametrize("B,H,W,T", [(24, 228, 304, 24), (3, 61, 77, 9), (2, 40, 150, 24)], ids=["config2", "w77", "w150"])
def test_training_step_without_the_tap_volume_equals_the_one_with_it(B, H, W, T, c_oracle, monkeypatch):
    """The default training path publishes S only (forward), rebuilds the taps in the reverse sweep and in the tail;
    CSPN_TRAIN_VOLUME=1 keeps the 8-plane volume.  Same output, same gradients, bit for bit — plain and with a sparse depth, also
    for widths that are not whole quads (the row-padding path, W_valid > 0, of the S-only forms)."""
    for sparse in (False, True):
        g, d, s = c_oracle.synthetic_inputs(230, B, H, W, 12, 500 if sparse else None)
        cot = dev(c_oracle.hash_normal(231, 9, (B, 1, H, W)))
        res = []
        for keep in ("0", "1"):
            monkeypatch.setattr(F, "_TRAIN_VOLUME", keep == "1")      # (the environment variable is read once, at import)
            gt, dt = dev(g).requires_grad_(True), dev(d).requires_grad_(True)
            out = pkg.CSPN_new.AffinityPropagate(T, 3)(gt, dt, dev(s) if sparse else None)
            out.backward(cot)
            res.append((out.detach(), gt.grad, dt.grad))
        for a, b in zip(*res):
            assert bits_equal(a, b)
    F.check_resident_errors()


class spin_limit(object):
    """Force the neighbour wait of every resident launch issued inside the block to give up after `n` polls."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.prev = F._RESIDENT_SPIN_LIMIT
        F._RESIDENT_SPIN_LIMIT = self.n

    def __exit__(self, *exc):
        F._RESIDENT_SPIN_LIMIT = self.prev
        return False


class guard(object):
    """Switch the device-side guard of plain resident inference calls (functional.set_resident_guard) inside the block."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.prev = F._RESIDENT_GUARD
        F.set_resident_guard(self.on)

    def __exit__(self, *exc):
        F.set_resident_guard(self.prev)
        return False


def _config2(c_oracle, seed=310, B=24):
    g, d, _ = c_oracle.synthetic_inputs(seed, B, 228, 304, 12, None)
    tg = np.abs(d + 0.1).astype(np.float32)
    return dev(g), dev(d), dev(tg)


def test_timeout_in_the_last_scored_forward_is_repaired_where_the_sums_are_used(c_oracle):
    """the LAST batch of an evaluation loop has no later resident launch to find its time-out.
    evaluation.all_gather_metric_sums / finalize_metrics wait for the pending launches and REPAIR the failed one before the
    sums are used: the refined depth is re-computed in place on the multi-launch schedule, the partial sums the failed launch
    added are taken out and the batch is scored exactly once — the totals equal an undisturbed loop's."""
    from cspn_monodepth_amd import evaluation as ev
    gt, dt, tg = _config2(c_oracle)
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    with torch.no_grad(), resident("on"):
        clean = ev.new_accumulator(DEV)
        for _ in range(4):
            ref = m.forward_scored(gt, dt, None, tg, clean)
        want, _ = ev.all_gather_metric_sums(clean)
        acc = ev.new_accumulator(DEV)
        for _ in range(3):
            m.forward_scored(gt, dt, None, tg, acc)
        with spin_limit(1):
            out = m.forward_scored(gt, dt, None, tg, acc)              # the last batch: nothing is launched after it
        total, _ = ev.all_gather_metric_sums(acc)                       # waits, repairs, THEN sums
        assert F.resident_fallbacks() == 1
        assert bits_equal(out, ref)                                    # repaired in place: the same bits
        assert np.allclose(total.cpu().numpy(), want.cpu().numpy(), rtol=1e-7)
        assert ev.finalize_metrics(total)["count"] == 4 * 24 * 228 * 304
        # ... and finalize_metrics on the device accumulator is a consumer of its own
        acc.zero_()
        with spin_limit(1):
            m.forward_scored(gt, dt, None, tg, acc)
        assert ev.finalize_metrics(acc)["count"] == 24 * 228 * 304 and F.resident_fallbacks() == 2
    F.check_resident_errors()


def test_three_timeouts_switch_auto_off_with_one_warning(c_oracle):
    """After _FALLBACK_LIMIT repaired time-outs mode "auto" turns itself off for the process (one RuntimeWarning says so; every
    earlier event has a RuntimeWarning of its own — ADVICE r4: no repair is silent): calls keep working on the multi-launch
    schedule, results unchanged."""
    import warnings
    gt, dt, _ = _config2(c_oracle, seed=312, B=3)
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    with torch.no_grad(), resident("auto"):
        with resident("off"):
            ref = m(gt, dt)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            outs = []
            for _ in range(F._FALLBACK_LIMIT):
                with spin_limit(1):
                    outs.append(m(gt, dt))
                F.ensure_resident_ok()
            assert F._RESIDENT_MODE == "off" and F.resident_fallbacks() == F._FALLBACK_LIMIT
            msgs = [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning)]
            assert len(msgs) == F._FALLBACK_LIMIT and len([t for t in msgs if "switched off" in t]) == 1
        assert F.resident_supported(gt, dt[:, 0], None, 24) is None     # the next call takes the multi-launch schedule
        assert all(bits_equal(o, ref) for o in outs) and bits_equal(m(gt, dt), ref)


def test_repair_touches_only_the_call_that_failed(c_oracle):
    """the repair re-runs exactly the journaled calls whose output holds the poison pattern of a tile that gave
    up.  An eval loop that refills static input buffers in place between batches must not get an earlier, CORRECT result
    overwritten with the refinement of a later batch's inputs; a failed call whose own inputs were overwritten before the time-out
    was noticed cannot be repaired and raises instead of producing another batch's result."""
    import warnings
    B, H, W, T = 3, 228, 304, 24
    ga, da, _ = c_oracle.synthetic_inputs(401, B, H, W, 12, None)
    gb, db, _ = c_oracle.synthetic_inputs(402, B, H, W, 12, None)
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    # (guard off: this is the HOST repair — what scored calls and hosts that switch the guard off rely on)
    with torch.no_grad(), resident("on"), guard(False), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with resident("off"):
            ref_a, ref_b = m(dev(ga), dev(da)), m(dev(gb), dev(db))
        g_buf, d_buf = dev(ga), dev(da)                     # the loop's static input buffers
        out_a = m(g_buf, d_buf)                             # batch A: clean
        torch.cuda.synchronize()
        g_buf.copy_(dev(gb)); d_buf.copy_(dev(db))          # refilled in place for batch B
        with spin_limit(1):
            out_b = m(g_buf, d_buf)                         # batch B: every tile gives up
        F.ensure_resident_ok()
        assert F.resident_fallbacks() == 1
        assert bits_equal(out_b, ref_b) and bits_equal(out_a, ref_a)       # B repaired, A left alone
        # a failed call whose inputs were overwritten before the host looked: not repairable, raised (and "on" stays on)
        n_c = F.resident_fallbacks()
        with spin_limit(1):
            out_c = m(g_buf, d_buf)
        torch.cuda.synchronize()
        if F.resident_fallbacks() == n_c:
            g_buf.copy_(dev(ga))
            with pytest.raises(F.ResidentLaunchTimeout, match="modified in place"):
                F.ensure_resident_ok()
            assert F._holds_poison(out_c)
        else:
            # the tiles gave up so quickly that the launch protocol's own look at the error word — it follows every launch — found it
            # inside the call and repaired the result at once, before this test could overwrite the inputs: a matter of timing (one run
            # in six on an idle box), and equally correct
            assert bits_equal(out_c, ref_b)
    F.check_resident_errors()


GUARD_SHAPES = [(24, 228, 304, 24), (3, 228, 304, 24), (1, 352, 1216, 24), (8, 352, 1216, 24), (2, 13, 20, 24), (5, 60, 64, 7),
                (3, 100, 148, 9), (1, 5, 4, 6), (2, 37, 8, 1), (4, 61, 76, 54)]


@pytest.mark.parametrize("B,H,W,T", GUARD_SHAPES, ids=["x".join(map(str, s)) for s in GUARD_SHAPES])
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_timed_out_result_is_complete_for_gpu_consumers(B, H, W, T, sparse, c_oracle):
    """the reference module is plain ATen — whatever consumes its output on the GPU sees the finished tensor
    (CSPN_new.py:80-92; in the reference's eval loop the consumer is Result.evaluate, single_gpu_trainer.py:129-137, not this
    package).  A plain inference call therefore carries a guard kernel behind its resident launch: every tile is forced to give
    up here (one poll), and a torch reduction enqueued right behind the call — no host check in between — sees the finished,
    bit-identical result, because the guard re-computed it on the stream."""
    import warnings
    g, d, s = c_oracle.synthetic_inputs(500 + B + T, B, H, W, 12, max(2, H * W // 140) if sparse else None)
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    gt, dt, st = dev(g), dev(d), dev(s)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with resident("off"):
            ref = m(gt, dt, st)
        with resident("on"), guard(True):
            n0 = F.resident_fallbacks()
            rp = F.resident_plan(B, H, W, T, int(sparse))
            multi_phase = rp["steps_per_phase"] < T and rp["tiles_x"] * rp["tiles_y"] > 1      # else: nobody to wait for
            with spin_limit(1):
                out = m(gt, dt, st)
            total = out.double().sum()                      # a GPU consumer this package knows nothing about
            nan_seen = torch.isnan(out).any()
            assert not bool(nan_seen) and float(total) == float(ref.double().sum())
            assert bits_equal(out, ref, T=T, sparse=sparse, which="guard-repaired")
            F.ensure_resident_ok()                          # the host still learns about the event (statistics / auto-off), no raise
            assert F.resident_fallbacks() == n0 + (1 if multi_phase else 0)     # (a single-phase launch waits for nobody)
    F.check_resident_errors()


@pytest.mark.parametrize("B,H,W", [(24, 228, 304), (3, 228, 304), (8, 352, 1216)], ids=["config2", "nyu_b3", "kitti_b8"])
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_scored_forward_under_guard_all(B, H, W, sparse, c_oracle):
    """set_resident_guard("all"): the scored forward carries the guard too — for a host that hands forward_scored's refined depth
    to other GPU work before it gathers the metrics.  Every tile gives up; the guard re-computes the depth (the multi-launch
    schedule's bits) and adds the metric terms of exactly the pixels the failed launch left unscored: a reduction enqueued right
    behind the call sees the finished depth, and the accumulated sums equal an undisturbed call's (to the summation order)."""
    import warnings
    from cspn_monodepth_amd import evaluation as ev
    T = 24
    g, d, s = c_oracle.synthetic_inputs(700 + B, B, H, W, 12, max(2, H * W // 140) if sparse else None)
    tgt = np.maximum(d + 0.1 * c_oracle.hash_normal(701, 9, d.shape), 0.0).astype(np.float32)
    tgt[c_oracle.hash_uniform(702, 9, d.shape) < 0.05] = 0.0
    gt, dt, st, tt = dev(g), dev(d), dev(s), dev(tgt)
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with resident("off"):
            acc0 = ev.new_accumulator(DEV)
            ref = m.forward_scored(gt, dt, st, tt, acc0)
        with resident("on"), guard("all"):
            acc1 = ev.new_accumulator(DEV)
            m.forward_scored(gt, dt, st, tt, acc1)              # a clean call first (registers the fast path)
            acc1.zero_()
            st_ = F._resident_state(gt.device)
            for how in ("general path", "fast path"):
                with spin_limit(1):                             # (a spin limit keeps the call on the general path ...)
                    out = m.forward_scored(gt, dt, st, tt, acc1) if how == "general path" else None
                if how == "fast path":                          # ... the fast path is forced to fail through its plan copy
                    fp = [v for v in F._SCORED_FAST.values() if (v.B, v.H, v.W) == (B, H, W) and v.blend == int(sparse)][0]
                    fp.plan_guarded.spin_limit = 1
                    try:
                        out = m.forward_scored(gt, dt, st, tt, acc1)
                    finally:
                        fp.plan_guarded.spin_limit = 0
                total = out.double().sum()
                assert float(total) == float(ref.double().sum()) and bits_equal(out, ref, T=T, sparse=sparse, which="scored, " + how)
                assert not st_["journal"]                       # guarded: nothing journaled
            assert np.allclose(acc1.sum(0).cpu().numpy(), 2 * acc0.sum(0).cpu().numpy(), rtol=1e-6)
            F.ensure_resident_ok()
            assert F.resident_fallbacks() >= 1
    F.check_resident_errors()


@pytest.mark.parametrize("name", golden_names("g1_") + golden_names("g2_"))
def test_guard_recomputation_on_reference_goldens(name):
    """The guard's re-computation against the vectors captured from the reference (small / degenerate / NaN-spreading /
    negative-sparse cases, odd widths through the row padding): every launch is forced to give up."""
    import warnings
    z = load_golden(name)
    g, d, s = z["guidance"], z["blur"], z.get("sparse")
    T = int(z["T"])
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    with torch.no_grad(), resident("on"), guard(True), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        with spin_limit(1):
            out = m(dev(g), dev(d), dev(s))
        got = out.cpu().numpy()
        try:
            F.ensure_resident_ok()
        except F.ResidentLaunchTimeout:
            pytest.fail("a guarded call must not raise")
    assert rel_err(got, z["out"]) <= 1e-5, name


def test_journal_does_not_pin_batches(c_oracle):
    """the journal of unchecked launches must not keep tens of batches of inputs alive.  Only the newest
    entries hold their tensors (the last two and what fits 256 MB); older ones are demoted to weak references, and entries
    behind a completed mark (one every 16 launches) are dropped on the next add: after 40 config-2-sized calls (80 MB each) at
    most 4 entries pin tensors; it was up to 32 batches = 2.5 GB."""
    gt, dt, _ = _config2(c_oracle, seed=411)
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    st = F._resident_state(gt.device)
    with torch.no_grad(), resident("on"):
        with guard(False):                          # host-repaired calls: their entries refer to the call's tensors
            for _ in range(40):
                m(gt, dt)
            j = st["journal"]
            assert len(j) <= F._JOURNAL_MAX
            strong = [e for e in j if not e.weak]
            assert len(strong) <= 4 and sum(e.nbytes for e in strong[:-2]) <= F._JOURNAL_STRONG_BYTES, (len(strong), len(j))
            torch.cuda.synchronize()
            m(gt, dt)
            assert len(st["journal"]) <= 16, len(st["journal"])     # everything behind the completed marks is gone
            F.ensure_resident_ok()
            assert not st["journal"]
        for _ in range(20):                         # guarded calls (the default for plain inference): nothing is kept at all
            m(gt, dt)
        assert not st["journal"] and st["guarded_pending"] == 20    # counted, not kept: no entry, no tensor reference
        F.ensure_resident_ok()


def test_contended_device_results_equal_uncontended(c_oracle):
    """a REAL co-tenant — a side-stream kernel holding 64 CUs (120 KB of LDS each: no resident workgroup fits
    beside it) for 5 ms at a time — while 50 scored forwards and 10 training steps run with a short neighbour wait (10 polls).
    Measured on MI355X (tools/probes/contention_probe.py): such a tenant does NOT dead-lock a 240-workgroup resident launch — the
    workgroups of the images that did get their CUs finish and free them for the rest, so the launch takes two or three launch
    times instead of one and the default wait of seconds (even 600 polls) rides it out; a wait of ~10 us does not.  Launches
    that give up are repaired; after three of them the process takes the multi-launch schedule.
    Inference never raises (refined depths and metric sums are repaired in place); a training step whose launch gave up raises a
    ResidentLaunchTimeout once and its retry succeeds; refined depths, metric sums and gradients equal the uncontended run's."""
    import ctypes
    from conftest import occupy_lib
    from cspn_monodepth_amd import evaluation as ev
    occ = occupy_lib()
    sink = torch.zeros(4, dtype=torch.int32, device=DEV)
    # four side streams, 16 tenant workgroups on each: HIP maps streams onto a handful of hardware queues round robin, and a
    # tenant that happens to share the launch stream's queue runs before or after the resident launches instead of beside them
    sides = [torch.cuda.Stream() for _ in range(4)]

    def tenant():
        for side in sides:
            assert occ.occupy(16, 120 * 1024, 500000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))
    gt, dt, tg = _config2(c_oracle, seed=313)
    m = pkg.CSPN_new.AffinityPropagate(24, 3)

    def run(contended):
        acc = ev.new_accumulator(DEV)
        outs, grads = [], []
        with torch.no_grad():
            for k in range(50):
                if contended and k % 5 == 0:
                    tenant()
                outs.append(m.forward_scored(gt, dt, None, tg, acc))
        total, _ = ev.all_gather_metric_sums(acc)
        for k in range(10):
            g_ = gt[:3].clone().requires_grad_(True)
            d_ = dt[:3].clone().requires_grad_(True)
            if contended:
                tenant()
            for attempt in range(2):
                try:
                    out = m(g_, d_, None)
                    (out * out).mean().backward()
                    break
                except F.ResidentLaunchTimeout:       # a training-form launch cannot be repaired after the fact: the step is
                    assert attempt == 0               # raised ONCE, mode "auto" is off by then, and the trainer's retry runs
                    g_.grad = d_.grad = None          # on the multi-launch schedule
            grads.append((g_.grad.clone(), d_.grad.clone()))
        torch.cuda.synchronize()
        return outs, total, grads

    with resident("auto"):
        ref_outs, ref_total, ref_grads = run(False)
        assert F.resident_fallbacks() == 0
        # 10 polls (~10 us) is shorter than the one or two launch times the tenant delays a workgroup by on the boxes measured; on a
        # box where it is not, the wait is shortened until launches do give up (1 poll always does), so that the repair is exercised
        # under the tenant whatever the timing
        for limit in (10, 3, 1):
            F.set_resident("auto")
            with spin_limit(limit):
                outs, total, grads = run(True)
            if F.resident_fallbacks() >= 1:
                break
        assert F.resident_fallbacks() >= 1                               # launches did give up, and were repaired
    for a, b_ in zip(outs, ref_outs):
        assert bits_equal(a, b_)
    # (the repaired batches are scored by the separate reduction — another summation order — and the failed launches' partial
    # sums are subtracted in fp64: equal to ~1e-9, not to the bit)
    assert np.allclose(total.cpu().numpy(), ref_total.cpu().numpy(), rtol=1e-7)
    for (ga, da), (gb, db) in zip(grads, ref_grads):
        assert torch.allclose(ga, gb, rtol=0, atol=1e-6 * float(gb.abs().max())) and torch.allclose(da, db, rtol=0, atol=1e-6 * float(db.abs().max()))
    F.ensure_resident_ok()


@pytest.mark.parametrize("where", ["forward", "reverse_sweep"])
@pytest.mark.parametrize("model", ["3x3", "k5_fp16"])
def test_timeout_in_a_training_step_raises_before_backward_returns(where, model, c_oracle):
    """WITHOUT the device-side guard (set_resident_guard(False), or a call beyond the guard's 54 halo pixels): a training forward
    (history kept) or the backward's reverse sweep that timed out must raise before `.backward()` returns — i.e. before any
    optimiser step could apply the gradients (functional._check_resident_at_end_of_backward).  The 3x3 model and the K = 5 fp16
    forms of config 3's shape (cspnk_forward_resident_history on the dot-product kernel, cspnk_transposed_resident)."""
    gt, dt, _ = _config2(c_oracle, seed=311, B=3)
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    if model == "k5_fp16":
        B, _, H, W = dt.shape
        gt = dev(c_oracle.hash_normal(312, 1, (B, 24, H, W))).half()
        dt = dt.half()
        m5 = pkg.CSPN_ours.AffinityPropagate(12, state_dtype=None)
        m = lambda g, d, s: m5(d, g, sparse_depth=s)      # noqa: E731
    with resident("on"), guard(False):
        for broken in (True, False):
            g_ = gt.clone().requires_grad_(True)
            d_ = dt.clone().requires_grad_(True)

            def step():
                with spin_limit(1 if (broken and where == "forward") else 0):
                    out = m(g_, d_, None)
                loss = (out * out).mean()
                with spin_limit(1 if (broken and where == "reverse_sweep") else 0):
                    loss.backward()
            if broken:
                # (raised by the end-of-backward check at the latest; a host that looks at the error word earlier — the launch
                #  protocol does after every launch — may raise from the forward call already)
                with pytest.raises(RuntimeError, match="timed out waiting for a neighbouring tile"):
                    step()
            else:
                step()                                                  # the path works again
                torch.cuda.synchronize()
                assert bool(torch.isfinite(g_.grad).all()) and bool(torch.isfinite(d_.grad).all())
    torch.cuda.synchronize()
    F.check_resident_errors()


@pytest.mark.parametrize("B,H,W", [(3, 228, 304), (24, 228, 304), (8, 352, 1216), (2, 61, 76)], ids=["nyu_b3", "config2", "kitti_b8", "small"])
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_guarded_training_step_survives_timeouts(B, H, W, sparse, c_oracle):
    """Round 5: the training forms carry the device-side guard as well (cspn_resident_plan.guard: the training forward and the
    volume-free reverse sweep have re-computation forms in csrc/cspn_repair.hip).  Every tile of the forward with history, then of
    the reverse sweep, is forced to give up: no exception, and the refined depth, the history planes the backward reads and both
    gradients are the bits of an undisturbed step — the loss and the optimiser never see a partial result, and the host does
    not wait at the end of the backward pass any more."""
    import warnings
    T = 24
    g, d, s = c_oracle.synthetic_inputs(600 + B, B, H, W, 12, max(2, H * W // 140) if sparse else None)
    cot = dev(c_oracle.hash_normal(601, 9, (B, 1, H, W)))
    m = pkg.CSPN_new.AffinityPropagate(T, 3)

    def step(limit_fwd, limit_bwd):
        g_, d_ = dev(g).requires_grad_(True), dev(d).requires_grad_(True)
        with spin_limit(limit_fwd):
            out = m(g_, d_, dev(s))
        with spin_limit(limit_bwd):
            out.backward(cot)
        return out.detach(), g_.grad, d_.grad

    with resident("on"), guard(True), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        ref = step(0, 0)
        torch.cuda.synchronize()
        assert F.resident_fallbacks() == 0
        for lf, lb in ((1, 0), (0, 1), (1, 1)):
            got = step(lf, lb)
            for a, b_, what in zip(got, ref, ("refined depth", "dL/dguidance", "dL/ddepth")):
                assert bits_equal(a, b_, T=T, sparse=sparse, which="%s, time-out forced in %s" % (what, "forward" if lf else "sweep"))
        F.ensure_resident_ok()                                  # the host learns about the events; nothing to raise
        assert F.resident_fallbacks() >= 1
    F.check_resident_errors()


def test_many_phases_repeated_on_one_workspace(c_oracle):
    """phase flags are seq + p + 1; the host used to advance seq by 8 per call, so with >= 10 phases (T >= 73 at
    8-step phases) the flags one call left behind satisfied the next call's first waits.  seq now advances by 256 and the
    engine refuses more than 255 phases.  T = 80 (10 phases) repeated on the cached workspace, alternating inputs."""
    T = 80
    sets = []
    for k in range(2):
        g, d, _ = c_oracle.synthetic_inputs(400 + k, 24, 228, 304, 12, None)
        sets.append((dev(g), dev(d)))
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    with torch.no_grad():
        with resident("off"):
            refs = [m(g, d) for g, d in sets]
        with resident("on"):
            assert F.resident_supported(sets[0][0], sets[0][1][:, 0], None, T) is not None
            for rep in range(12):
                for k in (0, 1, 1, 0):
                    assert bits_equal(m(*sets[k]), refs[k]), (rep, k)
    F.ensure_resident_ok()
    assert F._RES_SEQ_STEP >= 256
    with pytest.raises(RuntimeError, match="255 phases"):
        F.forward_resident(sets[0][0], sets[0][1][:, 0].contiguous(), None, 1100, 0, steps_per_phase=4)


@pytest.mark.parametrize("seed", range(8))
def test_random_shapes_resident_equals_multi_launch(seed, c_oracle):
    """Seeded sweep of batch / image size / step count / phase length (single-phase launches with halos deeper than a tile,
    regions that cannot be shifted into the image, odd step counts): resident = multi-launch, bit for bit, and the oracle."""
    rng = np.random.default_rng(5200 + seed)
    done = 0
    for _ in range(30):
        B, H, W = int(rng.integers(1, 7)), int(rng.integers(1, 100)), int(4 * rng.integers(1, 40))
        T, S = int(rng.integers(1, 26)), int(rng.choice([0, 4, 6, 8, 12]))
        sparse = bool(rng.random() < 0.5)
        if F.resident_plan(B, H, W, T, int(sparse), 0, S) is None:
            continue
        g, d, s = c_oracle.synthetic_inputs(900 + seed, B, H, W, 12, max(2, H * W // 100) if sparse else None)
        m = pkg.CSPN_new.AffinityPropagate(T, 3)
        with torch.no_grad():
            with resident("off"):
                ref = m(dev(g), dev(d), dev(s))
            out = F.forward_resident(dev(g), dev(d)[:, 0].contiguous(), None if s is None else dev(s)[:, 0].contiguous(), T, int(sparse),
                                     steps_per_phase=S)
        assert bits_equal(out, ref[:, 0]), (B, H, W, T, S, sparse)
        want = c_oracle.cspn3_forward(g, d, s, T)
        assert rel_err(out.cpu().numpy(), want[:, 0]) <= 1e-5, (B, H, W, T, S, sparse)
        done += 1
    F.ensure_resident_ok()
    assert done >= 8


def test_eval_under_inference_mode(c_oracle):
    """tensors made under torch.inference_mode() have no version counter; every unguarded resident launch — the default
    scored forward among them — builds a journal entry from its inputs and crashed there AFTER the kernel had been enqueued.  The eval
    loop's calls (plain, scored, sparse) under inference_mode return the bits they return under no_grad, and a forced time-out of the
    scored call is still repaired."""
    import warnings
    from cspn_monodepth_amd import evaluation as ev
    B, H, W, T = 3, 228, 304, 24
    g, d, s = c_oracle.synthetic_inputs(611, B, H, W, 12, 500)
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    with torch.no_grad():
        gt, dt, st = dev(g), dev(d), dev(s)
        tg = dt + 0.1
        acc0 = ev.new_accumulator("cuda")
        want_plain, want_scored = m(gt, dt, st).clone(), m.forward_scored(gt, dt, st, tg, acc0).clone()
        sums0, _ = ev.all_gather_metric_sums(acc0)
    with torch.inference_mode(), resident("on"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        gi, di, si = dev(g), dev(d), dev(s)                 # inference tensors
        ti = di + 0.1
        assert gi.is_inference()
        acc1 = ev.new_accumulator("cuda")
        assert bits_equal(m(gi, di, si), want_plain)
        assert bits_equal(m.forward_scored(gi, di, si, ti, acc1), want_scored)
        sums1, _ = ev.all_gather_metric_sums(acc1)
        assert torch.allclose(sums1, sums0, rtol=1e-12, atol=0)
        acc2 = ev.new_accumulator("cuda")
        with spin_limit(1):
            out = m.forward_scored(gi, di, si, ti, acc2)    # every tile gives up; repaired where the sums are used
            sums2, _ = ev.all_gather_metric_sums(acc2)
        assert bits_equal(out, want_scored) and torch.allclose(sums2, sums0, rtol=1e-6, atol=0)
    F.ensure_resident_ok()
    F.check_resident_errors()
