"""CPU tests: the oracle (numpy + C restatements) against the golden vectors captured from the
imported reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, rel_err, rmse
from oracle import cspn_oracle as orc

FWD_TOL = 2e-6      # oracle vs reference, fp32 (observed <= 9.6e-7, golden_manifest.json)


def _sp(z):
    return z.get("sparse")


@pytest.mark.parametrize("name", golden_names("g1_") + golden_names("g2_"))
def test_cspn3_forward_small(name, c_oracle):
    z = load_golden(name)
    T = int(z["T"])
    for impl in (orc, c_oracle):
        out = impl.cspn3_forward(z["guidance"], z["blur"], _sp(z), T)
        assert out.shape == z["out"].shape
        assert rel_err(out, z["out"]) <= FWD_TOL, (name, impl.__name__)


@pytest.mark.parametrize("name", golden_names("g3_") + golden_names("g4_"))
def test_cspn3_forward_full_frames(name, c_oracle):
    """Inputs regenerated from the integer-hash generator; output compared on the stored subsample + moments."""
    z = load_golden(name)
    _, H, W = (int(v) for v in z["shape"])
    ss = int(z["sparse_samples"])
    g, d, s = c_oracle.synthetic_inputs(int(z["seed"]), 1, H, W, 8, None if ss < 0 else ss)
    g2, d2, s2 = orc.synthetic_inputs(int(z["seed"]), 1, H, W, 8, None if ss < 0 else ss)
    assert np.array_equal(g, g2) and np.array_equal(d, d2) and (s is None or np.array_equal(s, s2))
    out = c_oracle.cspn3_forward(g, d, s, int(z["T"]))
    sub = int(z["sub"])
    assert rel_err(out[:, :, ::sub, ::sub], z["out_sub"]) <= FWD_TOL
    o64 = out.astype(np.float64)
    mom = np.array([o64.sum(), (o64 ** 2).sum()])
    assert np.allclose(mom, z["moments"], rtol=1e-6)


@pytest.mark.parametrize("name", golden_names("g5_"))
def test_cspn3_backward(name, c_oracle):
    z = load_golden(name)
    T = int(z["T"])
    f64 = name.endswith("f64")
    tol = 1e-10 if f64 else 2e-4
    for impl in (orc, c_oracle):
        gg, gd = impl.cspn3_backward(z["guidance"], z["blur"], _sp(z), z["cot"], T, np.float64)
        scale = max(1.0, float(np.abs(z["grad_guidance"]).max()))
        assert np.abs(gg - z["grad_guidance"]).max() <= tol * scale, name
        assert np.abs(gd - z["grad_blur"]).max() <= tol * max(1.0, float(np.abs(z["grad_blur"]).max())), name


@pytest.mark.parametrize("name", [n for n in golden_names("g6_") if "fp16" not in n])
def test_pac_forward_backward(name, c_oracle):
    z = load_golden(name)
    T = int(z["T"])
    for impl in (orc, c_oracle):
        out = impl.pac_forward(z["x"], z["guided"], _sp(z), T)
        assert rel_err(out, z["out"]) <= FWD_TOL, name
    gx, gg = orc.pac_backward(z["x"], z["guided"], _sp(z), z["cot"], T, np.float64)
    assert np.abs(gx - z["grad_x"]).max() <= 1e-10
    assert np.abs(gg - z["grad_guided"]).max() <= 1e-10


def test_pac_fp16_reference_quirk():
    """The reference promotes everything after the fp16 softmax to fp32 and returns fp32 (CSPN_ours.py:37)."""
    z = load_golden("g6_k5_t12_fp16")
    assert z["out"].dtype == np.float32
    kern16 = orc.pac_kernel(z["guided"].astype(np.float32))[0].astype(np.float16).astype(np.float32)
    # oracle fed with the fp16-rounded softmax reproduces the reference within fp16 softmax rounding
    out = orc.pac_forward(z["x"].astype(np.float32), z["guided"].astype(np.float32), None, int(z["T"]))
    assert rel_err(out, z["out"]) < 2e-3 and rmse(out, z["out"]) < 2e-3
    assert kern16.shape[1] == 25


@pytest.mark.parametrize("tag", ["nosp", "sp"])
def test_g15_reference_half_precision_runs_keep_their_recorded_distance(tag, c_oracle):
    """Golden G15 (round 6): the reference's CSPN_ours module on one full frame of fp16-rounded inputs, with fp16 taps + fp32 state and
    in half precision end to end.  The C oracle (fp32) on the same inputs reproduces the distances recorded when the fixture was made —
    the anchor the GPU test holds this package's fp16 paths to (tests/test_hip_kres.py)."""
    z = load_golden("g15_k5_t12_fp16_frame_%s" % tag)
    _, H, W = (int(v) for v in z["shape"])
    T, K, seed = int(z["T"]), int(z["K"]), int(z["seed"])
    gd = c_oracle.hash_normal(seed, 1, (1, K * K - 1, H, W)).astype(np.float16).astype(np.float32)
    x = c_oracle.hash_uniform(seed, 2, (1, 1, H, W), 0.0, 10.0).astype(np.float16).astype(np.float32)
    sp = c_oracle.hash_sparse(seed, 3, x, float(z["sparse_rate"])).astype(np.float16).astype(np.float32) if tag == "sp" else None
    want = c_oracle.pac_forward(x, gd, sp, T)
    scale = float(np.abs(want).max())
    assert z["out_half"].dtype == np.float16 and z["out_taps16"].dtype == np.float32
    for key, ek in (("out_taps16", "ref_err_taps16"), ("out_half", "ref_err_half")):
        r = z[key].astype(np.float32)
        assert abs(float(np.abs(r - want).max()) / scale - float(z[ek][0])) <= 1e-6
        assert abs(rmse(r, want) / scale - float(z[ek][1])) <= 1e-6
    assert float(z["ref_err_half"][0]) < 1e-3 and float(z["ref_err_taps16"][0]) < float(z["ref_err_half"][0])


def test_metrics_golden():
    z = load_golden("g7_metrics")
    got, n = orc.evaluate_metrics(z["pred"], z["target"])
    assert n == int((z["target"] > 0).sum())
    assert np.allclose(got, z["metrics"], rtol=2e-6)
    sums = orc.metric_sums(z["pred"], z["target"])
    assert sums[9] == n
    assert np.isclose(np.sqrt(sums[2] / n), z["metrics"][3], rtol=2e-6)


def test_unet_hook_golden(c_oracle):
    z = load_golden("g8_unet_hook")
    g, d, s = (z[k].astype(np.float32) for k in ("guidance_f16", "blur_f16", "sparse_f16"))
    out = c_oracle.cspn3_forward(g, d, s, int(z["T"]))
    assert np.abs(out - z["out"]).max() <= 1e-6 and rmse(out, z["out"]) <= 1e-7   # values are O(1e-2)


def test_hash_generator_c_equals_numpy(c_oracle):
    for shape in ((3, 5, 7), (1000,)):
        assert np.array_equal(orc.hash_uniform(7, 2, shape, -1.0, 3.0), c_oracle.hash_uniform(7, 2, shape, -1.0, 3.0))
        assert np.array_equal(orc.hash_normal(7, 1, shape), c_oracle.hash_normal(7, 1, shape))
    n = orc.hash_normal(3, 1, (200000,))
    assert abs(float(n.mean())) < 0.01 and abs(float(n.std()) - 1.0) < 0.01


def test_plumbing_port_matches_oracle():
    """The cpu_baseline op-mix port (bit-identical to the reference when generated) stays consistent."""
    import torch
    from oracle import ref_plumbing_torch as plumb
    g, d, s = orc.synthetic_inputs(5, 2, 12, 16, 12, 30)
    with torch.no_grad():
        out = plumb.cspn3_plumbing(torch.from_numpy(g), torch.from_numpy(d), torch.from_numpy(s), 7).numpy()
    assert rel_err(out, orc.cspn3_forward(g, d, s, 7)) <= FWD_TOL
    z = load_golden("g1_c12_t24_sp")
    with torch.no_grad():
        out = plumb.cspn3_plumbing(*(torch.from_numpy(z[k]) for k in ("guidance", "blur", "sparse")), 24).numpy()
    assert rel_err(out, z["out"]) <= FWD_TOL
    gd = orc.hash_normal(1, 1, (1, 24, 9, 10)); x = orc.hash_uniform(1, 2, (1, 1, 9, 10), 0, 10)
    with torch.no_grad():
        out = plumb.pac_plumbing(torch.from_numpy(x), torch.from_numpy(gd), None, 5).numpy()
    assert rel_err(out, orc.pac_forward(x, gd, None, 5)) <= FWD_TOL


@pytest.mark.parametrize("name", golden_names("g11_"))
def test_pac_multichannel_golden(name):
    """x [B,C>1,H,W] through CSPN_ours (CSPN_ours.py:24-29; pac.py:89-92 broadcast): forward from the module, gradients
    from the reference's autograd through pac.conv2d(native_impl=True)."""
    z = load_golden(name)
    T = int(z["T"])
    assert z["x"].shape[1] > 1 and z["out"].shape == z["x"].shape
    out = orc.pac_forward_multichannel(z["x"], z["guided"], _sp(z), T)
    assert rel_err(out, z["out"]) <= FWD_TOL, name
    gx, gg = orc.pac_backward_multichannel(z["x"], z["guided"], _sp(z), z["cot"], T, np.float64)
    assert np.abs(gx - z["grad_x"]).max() <= 1e-10 and np.abs(gg - z["grad_guided"]).max() <= 1e-10
