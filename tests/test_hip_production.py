"""GPU parity at PRODUCTION size on the DEFAULT plan: the kernel instances bench.py times are the
ones compared with the oracle here — forward and hand-written backward of BASELINE configs 2, 3, 4 and of the per-GPU
shards of configs 4 / 5 (KITTI B=1, NYU B=3).  Every case first asserts which instance the built-in heuristic picks, so
a heuristic change cannot silently move the test off the production path.

Reference: network/libs/post_process/CSPN_new.py:80-92 (3x3 loop), CSPN_ours.py:47-53 (K x K loop); gradients =
the autograd graph of the same lines (closed form of SURVEY.md §3.2, restated in oracle/)."""
import numpy as np
import pytest
import torch

import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
from conftest import bits_equal, rel_err, rmse
from oracle import cspn_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL_TOL, RMSE_TOL = 1e-5, 1e-4          # BASELINE.json north_star


def dev(x, grad=False):
    if x is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t.requires_grad_(True) if grad else t


def grad_close(got, want, tol):
    return float(np.abs(got - want).max()) <= tol * max(1.0, float(np.abs(want).max()))


# (label, B, H, W, expected threads of the multi-launch forward instance, expected steps per launch)
CASES3 = [
    ("config2_nyu_b24", 24, 228, 304, 1024, 8),
    ("config5_shard_nyu_b3", 3, 228, 304, 512, 8),
    ("config4_shard_kitti_b1", 1, 352, 1216, 1024, 8),
    ("config4_kitti_b8", 8, 352, 1216, 1024, 8),
]
# The instance the DEFAULT path really runs: under no-grad the module takes cspn3_resident (MODE 0 / 1), under
# grad its training forms (MODE 2 forward with history, MODE 4 reverse sweep) — all on the tiling cspn3_resident_plan finds for the
# shape on the device's CUs.  label -> (steps per phase, tiles_x, tiles_y, tile_w, tile_h, quads per thread = the NQ template
# argument, images per launch, launches) on the 256 CUs of an MI355X; a change to resident_geometry (cspn_resident.hip) that moves
# a production shape to another instance must show up here.
RESIDENT3 = {
    "config2_nyu_b24": (8, 2, 5, 152, 46, 5, 24, 1),
    "config5_shard_nyu_b3": (8, 7, 12, 44, 19, 1, 3, 1),
    "config4_shard_kitti_b1": (8, 11, 22, 112, 16, 2, 1, 1),
    "config4_kitti_b8": (8, 8, 8, 152, 44, 5, 4, 2),
}
_RP_KEYS = ("steps_per_phase", "tiles_x", "tiles_y", "tile_w", "tile_h", "quads_per_thread", "images_per_launch", "launches")


def assert_resident_instance(rp, want, label):
    assert torch.cuda.get_device_properties(0).multi_processor_count == 256, "the expected tilings are those of an MI355X (256 CUs)"
    assert rp is not None, label
    assert tuple(rp[k] for k in _RP_KEYS) == want and rp["threads"] in (512, 768, 256, 1024), (label, {k: rp[k] for k in _RP_KEYS})


@pytest.mark.parametrize("label,B,H,W,threads,S", CASES3, ids=[c[0] for c in CASES3])
@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_cspn3_forward_backward_production(label, B, H, W, threads, S, sparse, c_oracle):
    T = 24
    for hist in (False, True):                                         # inference plan and the history-keeping plan
        p = F.resolve_plan(3, B, H, W, T, hist, None)
        assert (p["threads"], p["steps_per_launch"], p["quads_per_thread"], p["force_scalar"]) == (threads, S, 1, 0), (label, p)
    assert (1, threads) in F._FROM_GUIDANCE_INSTANCES and (1, threads) in F._TRANSPOSED_INSTANCES[3]
    g, d, s = c_oracle.synthetic_inputs(7, B, H, W, 12, 500 if sparse else None)
    cot = c_oracle.hash_normal(8, 9, (B, 1, H, W))
    want = c_oracle.cspn3_forward(g, d, s, T)
    wg, wd = c_oracle.cspn3_backward(g, d, s, cot, T, np.float64)

    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    gt, dt, st = dev(g, True), dev(d, True), dev(s)
    assert F.from_guidance_supported(gt, dt[:, 0], None if st is None else st[:, 0])   # the fused-prepare entry runs
    # ... and the weight-resident launches serve both the no-grad call and the training forms, on this tiling
    assert F._RESIDENT_MODE == "auto"
    rp = F.resident_supported(gt.detach(), dt.detach()[:, 0], None if st is None else st[:, 0], T)
    assert_resident_instance(rp, RESIDENT3[label], label)
    assert rp["threads"] == 512
    n_res = F._resident_state(gt.device)["seq"]
    out = m(gt, dt, st)
    out.backward(dev(cot))
    torch.cuda.synchronize()
    o = out.detach().cpu().numpy()
    assert rel_err(o, want) <= REL_TOL and rmse(o, want) <= RMSE_TOL, label
    assert grad_close(gt.grad.cpu().numpy(), wg, 5e-4), label
    assert grad_close(dt.grad.cpu().numpy(), wd, 5e-5), label
    assert torch.count_nonzero(gt.grad[:, 8:]) == 0
    with torch.no_grad():                                              # the no-grad entry (what bench.py's step runs)
        o2 = m(gt.detach(), dt.detach(), st)
    assert bits_equal(o2, out.detach(), T=T, sparse=sparse)
    # three resident calls were issued: training forward (MODE 2), reverse sweep (MODE 4), no-grad forward (MODE 0)
    assert F._resident_state(gt.device)["seq"] - n_res == 3 * F._RES_SEQ_STEP and F.resident_fallbacks() == 0


def _pac_inputs(c_oracle, B, H, W, K, sparse):
    gd = c_oracle.hash_normal(40 + K, 1, (B, K * K - 1, H, W))
    x = c_oracle.hash_uniform(40 + K, 2, (B, 1, H, W), 0.0, 10.0)
    s = c_oracle.hash_sparse(40 + K, 3, x, 500.0 / (H * W)) if sparse else None
    return x, gd, s


@pytest.mark.parametrize("sparse", [False, True], ids=["nosparse", "sparse"])
def test_config3_full_size_fp16(sparse, c_oracle):
    """BASELINE config 3: B=24, 228x304, 5x5 softmax affinity, 12 steps, fp16 — the pair-interleaved fp16 tap volume
    on its production plan (S=4, NQ=3, 256 threads), against the fp32 oracle on the fp16-rounded inputs, in both
    state modes (fp16 storage: what bench.py --workload pac5 times; "reference": fp32 state, CSPN_ours.py:37)."""
    B, H, W, K, T = 24, 228, 304, 5, 12
    p = F.resolve_plan(K, B, H, W, T, False, F.dtype_default_plan(K, torch.float16, None))
    assert (p["steps_per_launch"], p["quads_per_thread"], p["threads"], p["force_scalar"]) == (4, 3, 256, 0), p
    x, gd, s = _pac_inputs(c_oracle, B, H, W, K, sparse)
    x16, gd16 = x.astype(np.float16), gd.astype(np.float16)
    s16 = None if s is None else s.astype(np.float16)
    # the instance the default (no-grad) call really runs: cspnk_d2<BLEND, MODE, CLEAN, 768> — ONE launch of 2 x 12 images on a
    # 2 x 10 tiling with 4-step phases, the dot-product step form (fp16 planes: state_dtype None); fp32 planes ("reference")
    # stay on the FMA kernel cspnk_resident with the same tiling, two launches
    kp = F.pac_resident_supported(dev(gd16), dev(x16)[:, 0], None if s16 is None else dev(s16)[:, 0], T)
    assert_resident_instance(kp, (4, 2, 10, 152, 23, 1, 12, 2), "config3")
    assert kp["threads"] == 768 and F._KRES_STEP_FORM == F._lib.STEP_AUTO
    f32 = lambda a: None if a is None else a.astype(np.float32)       # noqa: E731
    want = c_oracle.pac_forward(f32(x16), f32(gd16), f32(s16), T)
    scale = float(np.abs(want).max())
    for state, tol_max, tol_rmse in (("reference", 4e-3, 1e-3), (None, 8e-3, 3e-3)):
        with torch.no_grad():
            out = pkg.CSPN_ours.AffinityPropagate(T, state_dtype=state)(dev(x16), dev(gd16), sparse_depth=dev(s16))
        assert out.dtype == (torch.float32 if state == "reference" else torch.float16)
        o = out.float().cpu().numpy()
        assert np.isfinite(o).all()
        assert float(np.abs(o - want).max()) <= tol_max * scale, (state, float(np.abs(o - want).max()))
        assert rmse(o, want) <= tol_rmse * scale, (state, rmse(o, want))
    # fp32 module on the same full-size problem: the north-star tolerance
    with torch.no_grad():
        o32 = pkg.CSPN_ours.AffinityPropagate(T)(dev(x), dev(gd), sparse_depth=dev(s)).cpu().numpy()
    w32 = c_oracle.pac_forward(x, gd, s, T)
    assert rel_err(o32, w32) <= REL_TOL and rmse(o32, w32) <= RMSE_TOL


def test_config3_backward_full_size(c_oracle):
    """K=5 hand-written backward at B=24, 228x304, T=12 (fp32) against the fp64 numpy restatement of the autograd of
    pac.py:124-144 (native_impl branch) — the reverse sweep + fused tail instances bench.py's training leg times."""
    B, H, W, K, T = 24, 228, 304, 5, 12
    x, gd, s = _pac_inputs(c_oracle, B, H, W, K, True)
    cot = c_oracle.hash_normal(49, 9, (B, 1, H, W))
    wx, wgd = orc.pac_backward(x, gd, s, cot, T, np.float64)
    xt, gdt = dev(x, True), dev(gd, True)
    out = pkg.CSPN_ours.AffinityPropagate(T)(xt, gdt, sparse_depth=dev(s))
    out.backward(dev(cot))
    assert grad_close(xt.grad.cpu().numpy(), wx, 5e-5)
    assert grad_close(gdt.grad.cpu().numpy(), wgd, 5e-4)


def test_scored_forward_production(c_oracle):
    """forward_scored at config 2 (the exact call bench.py's step makes): refined depth equals the plain forward bit for
    bit and the fused metric sums equal the numpy oracle's on the oracle's refined depth."""
    B, H, W, T = 24, 228, 304, 24
    g, d, s = c_oracle.synthetic_inputs(9, B, H, W, 12, 500)
    tgt = np.maximum(d + 0.1 * c_oracle.hash_normal(10, 9, d.shape), 0.0).astype(np.float32)
    tgt[c_oracle.hash_uniform(11, 9, d.shape) < 0.05] = 0.0
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    for sp in (None, s):
        acc = pkg.evaluation.new_accumulator(DEV)
        with torch.no_grad():
            out = m.forward_scored(dev(g), dev(d), dev(sp), dev(tgt), acc)
            ref = m(dev(g), dev(d), dev(sp))
        assert bits_equal(out, ref)
        want = orc.metric_sums(c_oracle.cspn3_forward(g, d, sp, T), tgt)
        assert np.allclose(acc.sum(0).cpu().numpy(), want, rtol=2e-5)
