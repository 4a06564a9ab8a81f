"""GPU tests of the hand-written backward (closed form of SURVEY.md §3.2) against gradients captured
from the reference's autograd (goldens G5/G6) and against the C oracle at larger sizes."""
import numpy as np
import pytest
import torch

import cspn_monodepth_amd as pkg
from conftest import golden_names, load_golden
from oracle import cspn_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x, grad=False):
    if x is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t.requires_grad_(True) if grad else t


def close(got, want, tol):
    scale = max(1.0, float(np.abs(want).max()))
    return float(np.abs(got - want).max()) <= tol * scale


@pytest.mark.parametrize("name", [n for n in golden_names("g5_") if n.endswith("f32")])
def test_cspn3_grad_golden(name):
    z = load_golden(name)
    g, d = dev(z["guidance"], True), dev(z["blur"], True)
    out = pkg.CSPN_new.AffinityPropagate(int(z["T"]), 3)(g, d, dev(z.get("sparse")))
    out.backward(dev(z["cot"]))
    z64 = load_golden(name[:-3] + "f64")           # fp64 reference gradient = ground truth
    assert close(g.grad.cpu().numpy(), z64["grad_guidance"], 2e-4), name
    assert close(d.grad.cpu().numpy(), z64["grad_blur"], 2e-5), name
    assert g.grad.shape == g.shape and d.grad.shape == d.shape
    if g.shape[1] > 8:
        assert torch.count_nonzero(g.grad[:, 8:]) == 0       # unused head channels get exactly zero


@pytest.mark.parametrize("plan", [None, dict(steps_per_launch=3, tile_w=32, tile_h=28, quads_per_thread=2, threads=256)])
@pytest.mark.parametrize("sparse", [False, True])
def test_cspn3_grad_vs_oracle(plan, sparse, c_oracle):
    B, H, W, T = 2, 44, 60, 9
    g, d, s = c_oracle.synthetic_inputs(31, B, H, W, 12, 120 if sparse else None)
    cot = c_oracle.hash_normal(32, 9, (B, 1, H, W))
    wg, wd = c_oracle.cspn3_backward(g, d, s, cot, T, np.float64)
    gt, dt = dev(g, True), dev(d, True)
    out = pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)(gt, dt, dev(s))
    out.backward(dev(cot))
    assert close(gt.grad.cpu().numpy(), wg, 5e-4) and close(dt.grad.cpu().numpy(), wd, 5e-5)
    # forward under autograd (history kept) equals the no-grad forward
    with torch.no_grad():
        out2 = pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)(dev(g), dev(d), dev(s))
    assert torch.equal(out.detach(), out2)


def test_cspn3_grad_only_depth(c_oracle):
    g, d, s = c_oracle.synthetic_inputs(33, 1, 16, 20, 8, 30)
    cot = c_oracle.hash_normal(34, 9, (1, 1, 16, 20))
    _, wd = c_oracle.cspn3_backward(g, d, s, cot, 4, np.float64)
    dt = dev(d, True)
    pkg.CSPN_new.AffinityPropagate(4, 3)(dev(g), dt, dev(s)).backward(dev(cot))
    assert close(dt.grad.cpu().numpy(), wd, 5e-5)


@pytest.mark.parametrize("name", [n for n in golden_names("g6_") if "fp16" not in n])
def test_pac_grad_golden(name):
    z = load_golden(name)
    x, gd = dev(z["x"], True), dev(z["guided"], True)
    out = pkg.CSPN_ours.AffinityPropagate(int(z["T"]))(x, gd, sparse_depth=dev(z.get("sparse")))
    out.backward(dev(z["cot"]))
    assert close(x.grad.cpu().numpy(), z["grad_x"], 5e-5), name
    assert close(gd.grad.cpu().numpy(), z["grad_guided"], 5e-4), name


@pytest.mark.parametrize("sparse", [False, True])
def test_fused_tail_equals_unfused_path(sparse, c_oracle):
    """cspn3_backward_tail / cspn_pac_backward_tail (one pass, dL/dw kept in registers) against the three-call
    form (cspn_grad_weights + guidance / softmax kernels) on the same histories, incl. image-border tiles."""
    from cspn_monodepth_amd import functional as F
    B, H, W, T = 2, 23, 36, 5
    g, d, s = c_oracle.synthetic_inputs(41, B, H, W, 12, 60 if sparse else None)
    cot = dev(c_oracle.hash_normal(42, 9, (B, 1, H, W)))
    sp = dev(s)[:, 0].contiguous() if sparse else None
    gt, d0 = dev(g), dev(d)[:, 0].contiguous()
    w8, S, _ = F.cspn3_prepare(gt, want_s=True)
    _, hist = F.propagate(w8, d0, sp, 3, T, F.BLEND_SPARSE if sparse else F.BLEND_NONE, keep_history=True)
    g_T, ghist = F._reverse_sweep(w8, 3, T, sp, cot, None)
    gw, gd0_ref = F._grad_weights(w8, 3, T, d0, hist, sp, g_T, ghist)
    L, P, st = F._lib.lib(), F._p, F._stream(gt.device)
    gg_ref, gg = torch.empty_like(gt), torch.full_like(gt, float("nan"))
    gd0 = torch.empty_like(gd0_ref)
    assert L.cspn3_grad_guidance(P(gt), 0, gt.stride(0), gt.stride(1), 12, P(w8), 0, P(S), P(gw), P(gg_ref), B, H, W, st)
    assert L.cspn3_backward_tail(P(d0), P(hist), P(g_T), P(ghist), P(sp), P(gt), gt.stride(0), gt.stride(1), 12, P(w8), P(S),
                                 P(gg), P(gd0), 0, B, H, W, T, st)
    torch.cuda.synchronize()
    assert not torch.isnan(gg).any()                      # every element written (zero-fill of border targets)
    assert torch.allclose(gg, gg_ref, rtol=1e-5, atol=1e-6 * float(gg_ref.abs().max()))
    assert torch.allclose(gd0, gd0_ref, rtol=1e-5, atol=1e-6)
    # K x K
    for K in (3, 5):
        gd = dev(c_oracle.hash_normal(43, 1, (B, K * K - 1, H, W)))
        wk, _ = F.pac_prepare(gd)
        _, hist = F.propagate(wk, d0, sp, K, T, F.BLEND_SPARSE if sparse else F.BLEND_NONE, keep_history=True)
        g_T, ghist = F._reverse_sweep(wk, K, T, sp, cot, None)
        gw, gx_ref = F._grad_weights(wk, K, T, d0, hist, sp, g_T, ghist)
        ref, out = torch.empty_like(gd), torch.full_like(gd, float("nan"))
        gx = torch.empty_like(gx_ref)
        assert L.cspn_pac_grad_guided(P(wk), 0, P(gw), P(ref), 0, B, H, W, K, st)
        assert L.cspn_pac_backward_tail(P(d0), P(hist), P(g_T), P(ghist), P(sp), P(wk), P(out), P(gx), 0, 0, 0, B, H, W, K, T, st)
        torch.cuda.synchronize()
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))
        assert torch.allclose(gx, gx_ref, rtol=1e-5, atol=1e-6)


def test_fp16_backward_runs_and_tracks_fp32(c_oracle):
    """fp16 storage (pair-interleaved tap volume) through the whole backward: loose agreement with fp64 oracle."""
    B, H, W, T = 2, 24, 36, 6
    g, d, s = c_oracle.synthetic_inputs(51, B, H, W, 12, 40)
    cot = c_oracle.hash_normal(52, 9, (B, 1, H, W))
    g16, d16, s16 = (a.astype(np.float16) for a in (g, d, s))
    wg, wd = c_oracle.cspn3_backward(g16.astype(np.float32), d16.astype(np.float32), s16.astype(np.float32), cot, T, np.float64)
    gt, dt = dev(g16, True), dev(d16, True)
    out = pkg.CSPN_new.AffinityPropagate(T, 3)(gt, dt, dev(s16))
    out.backward(dev(cot.astype(np.float16)))
    assert gt.grad.dtype == torch.float16 and dt.grad.dtype == torch.float16
    assert close(gt.grad.float().cpu().numpy(), wg, 3e-2) and close(dt.grad.float().cpu().numpy(), wd, 1e-2)
    # K x K, fp16 guided with fp32 state (the reference's promotion) and fp16 state
    K = 5
    gd = c_oracle.hash_normal(53, 1, (B, 24, H, W)).astype(np.float16)
    x = c_oracle.hash_uniform(53, 2, (B, 1, H, W), 0.0, 10.0).astype(np.float16)
    wx, wgd = orc.pac_backward(x.astype(np.float32), gd.astype(np.float32), None, cot, T, np.float64)
    for state in ("reference", None):
        xt, gdt = dev(x, True), dev(gd, True)
        o = pkg.CSPN_ours.AffinityPropagate(T, state_dtype=state)(xt, gdt)
        o.backward(dev(cot.astype(np.float32 if state == "reference" else np.float16)))
        assert close(xt.grad.float().cpu().numpy(), wx, 1e-2) and close(gdt.grad.float().cpu().numpy(), wgd, 3e-2)


@pytest.mark.parametrize("name", golden_names("g11_"))
def test_pac_multichannel_golden(name):
    """CSPN_ours with x [B,C>1,H,W] (VERDICT r01 missing #3 / ADVICE): forward vs the reference module, gradients vs the
    reference's autograd (goldens G11)."""
    z = load_golden(name)
    x, gd = dev(z["x"], True), dev(z["guided"], True)
    out = pkg.CSPN_ours.AffinityPropagate(int(z["T"]))(x, gd, sparse_depth=dev(z.get("sparse")))
    assert out.shape == x.shape
    o = out.detach().cpu().numpy()
    assert float((np.abs(o - z["out"]) / np.maximum(np.abs(z["out"]), 1e-6)).max()) <= 1e-5, name
    out.backward(dev(z["cot"]))
    assert close(x.grad.cpu().numpy(), z["grad_x"], 5e-5), name
    assert close(gd.grad.cpu().numpy(), z["grad_guided"], 5e-4), name
    with torch.no_grad():
        o2 = pkg.CSPN_ours.AffinityPropagate(int(z["T"]))(x.detach(), gd.detach(), sparse_depth=dev(z.get("sparse")))
    assert torch.equal(o2, out.detach())
