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
