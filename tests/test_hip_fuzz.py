"""Seeded random sweep of shapes / step counts / plans / variants against the C oracle (GPU).  Catches geometry
corner cases (tiles vs image borders, halo vs tiny images, remainder launches, row padding) that the hand-picked
cases may miss."""
import numpy as np
import pytest
import torch

import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _random_plan(rng, K, B, H, W, T):
    """A random plan that resolves (or None for the built-in heuristic)."""
    if rng.random() < 0.35:
        return None
    for _ in range(20):
        nq, threads = [(1, 256), (2, 256), (4, 256), (1, 512), (2, 512), (1, 1024), (2, 1024)][rng.integers(0, 7)] \
            if K == 3 else ([(1, 256), (2, 256), (1, 512)][rng.integers(0, 3)] if K == 5 else (1, 256))
        plan = dict(steps_per_launch=int(rng.integers(1, min(T, 9) + 1)), tile_w=int(4 * rng.integers(1, 24)),
                    tile_h=int(rng.integers(1, 70)), quads_per_thread=nq, threads=threads)
        try:
            F.resolve_plan(K, B, H, -(-W // 4) * 4, T, False, plan)
            return plan
        except RuntimeError:
            continue
    return None


@pytest.mark.parametrize("seed", range(12))
def test_random_cspn3_cases(seed, c_oracle):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(6):
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 90)), int(rng.integers(1, 150))
        T = int(rng.integers(1, 26))
        C = int(rng.choice([8, 12]))
        sparse = rng.random() < 0.5
        g, d, s = c_oracle.synthetic_inputs(int(rng.integers(0, 1 << 30)), B, H, W, C, max(1, H * W // 30) if sparse else None)
        if rng.random() < 0.3 and H > 2 and W > 2:      # an all-zero-gate pixel: 0/0 = NaN that spreads
            y, x = int(rng.integers(1, H - 1)), int(rng.integers(1, W - 1))
            for k, (dy, dx) in enumerate(((1, 1), (1, 0), (1, -1), (0, 1), (0, -1), (-1, 1), (-1, 0), (-1, -1))):
                g[0, k, y + dy, x + dx] = 0.0
        plan = _random_plan(rng, 3, B, H, W, T)
        want = c_oracle.cspn3_forward(g, d, s, T)
        with torch.no_grad():
            out = pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)(dev(g), dev(d), dev(s))
        assert rel_err(out.cpu().numpy(), want) <= 1e-5, (B, H, W, T, C, sparse, plan)


@pytest.mark.parametrize("seed", range(6))
def test_random_pac_cases(seed, c_oracle):
    rng = np.random.default_rng(2000 + seed)
    for _ in range(4):
        K = int(rng.choice([3, 5, 7]))
        B, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 60)), int(rng.integers(1, 90))
        T = int(rng.integers(1, 13))
        gd = c_oracle.hash_normal(int(rng.integers(0, 1 << 30)), 1, (B, K * K - 1, H, W))
        x = c_oracle.hash_uniform(int(rng.integers(0, 1 << 30)), 2, (B, 1, H, W), 0.0, 10.0)
        s = c_oracle.hash_sparse(7, 3, x, 0.05) if rng.random() < 0.5 else None
        plan = _random_plan(rng, K, B, H, W, T)
        want = c_oracle.pac_forward(x, gd, s, T)
        with torch.no_grad():
            out = pkg.CSPN_ours.AffinityPropagate(T, plan=plan)(dev(x), dev(gd), sparse_depth=dev(s))
        assert rel_err(out.cpu().numpy(), want) <= 1e-5, (K, B, H, W, T, plan)


@pytest.mark.parametrize("seed", range(6))
def test_random_gradient_cases(seed, c_oracle):
    rng = np.random.default_rng(3000 + seed)
    for _ in range(3):
        B, H, W, T = int(rng.integers(1, 3)), int(rng.integers(2, 40)), int(rng.integers(2, 70)), int(rng.integers(1, 10))
        sparse = rng.random() < 0.5
        g, d, s = c_oracle.synthetic_inputs(int(rng.integers(0, 1 << 30)), B, H, W, 12, max(1, H * W // 30) if sparse else None)
        cot = c_oracle.hash_normal(int(rng.integers(0, 1 << 30)), 9, (B, 1, H, W))
        wg, wd = c_oracle.cspn3_backward(g, d, s, cot, T, np.float64)
        gt, dt = dev(g).requires_grad_(True), dev(d).requires_grad_(True)
        plan = _random_plan(rng, 3, B, H, W, T)
        pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)(gt, dt, dev(s)).backward(dev(cot))
        sg, sd = max(1.0, float(np.abs(wg).max())), max(1.0, float(np.abs(wd).max()))
        assert float(np.abs(gt.grad.cpu().numpy() - wg).max()) <= 1e-3 * sg, (B, H, W, T, sparse, plan)
        assert float(np.abs(dt.grad.cpu().numpy() - wd).max()) <= 1e-4 * sd, (B, H, W, T, sparse, plan)


def test_special_values_in_depth(c_oracle):
    """inf / nan / huge / tiny depths: non-finite patterns must spread exactly as in the reference arithmetic
    (0 * inf = nan, inf - inf = nan, ...), also under temporal blocking and from-guidance weights."""
    rng = np.random.default_rng(5)
    B, H, W, T = 2, 40, 56, 7
    g, d, s = c_oracle.synthetic_inputs(77, B, H, W, 12, 60)
    flat = d.reshape(-1)
    idx = rng.choice(flat.size, 12, replace=False)
    # (values within 8x of FLT_MAX are left out: the reference overflows in its un-normalised sum  sum_k A_k d_k  where
    #  the engine, which multiplies by pre-normalised weights <= 1, does not — DESIGN.md §1)
    flat[idx] = np.array([np.inf, -np.inf, np.nan, 1e36, -1e36, 1e-38, 1e-45, 0.0, np.inf, np.nan, 1e30, -1e30], np.float32)
    g[0, 3, 5, 7] = 0.0
    g[1, :, 20, 30] = 0.0
    for sp in (None, s):
        want = c_oracle.cspn3_forward(g, d, sp, T)
        for plan in (None, dict(steps_per_launch=1, tile_w=32, tile_h=32, quads_per_thread=1, threads=256),
                     dict(steps_per_launch=3, tile_w=24, tile_h=26, quads_per_thread=2, threads=256)):
            with torch.no_grad():
                out = pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)(dev(g), dev(d), dev(sp)).cpu().numpy()
            assert np.array_equal(np.isnan(out), np.isnan(want)), plan
            assert np.array_equal(np.isposinf(out), np.isposinf(want)) and np.array_equal(np.isneginf(out), np.isneginf(want))
            fin = np.isfinite(want)
            assert np.allclose(out[fin], want[fin], rtol=1e-5, atol=1e-30)
