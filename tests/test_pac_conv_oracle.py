"""CPU tests for the general pixel-adaptive conv op: the numpy oracle against the golden vectors
captured from the reference (tests/golden/g9_*.npz, made by make_golden_pac_conv.py), and the host-side mirror of
network/libs/base/pac.py (shapes, errors, no CPU fallback) without touching a GPU."""
import inspect

import numpy as np
import pytest
import torch

from cspn_monodepth_amd.base import pac
from conftest import golden_names, load_golden
from oracle import pac_oracle as porc

CONV_CASES = [n for n in golden_names("g9_") if "nd2col" not in n and n != "g9_fp16"]


def geom_of(z):
    kh, kw, sh, sw, ph, pw, dh, dw = (int(v) for v in z["geom"][:8])
    return (kh, kw), (sh, sw), (ph, pw), (dh, dw)


def nmax(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


@pytest.mark.parametrize("name", CONV_CASES)
def test_oracle_forward_and_gradients_match_reference(name):
    z = load_golden(name)
    k, s, p, d = geom_of(z)
    out = porc.pac_conv2d_forward(z["x"], z["kernel"], k, s, p, d)
    assert out.shape == z["out"].shape and nmax(out, z["out"]) < 1e-6
    gi, gk = porc.pac_conv2d_backward(z["x"], z["kernel"], z["cot"], k, s, p, d)
    assert gi.shape == z["grad_input_f64"].shape and gk.shape == z["grad_kernel_f64"].shape
    assert np.abs(gi - z["grad_input_f64"]).max() < 1e-12
    assert np.abs(gk - z["grad_kernel_f64"]).max() < 1e-12
    gi32, gk32 = porc.pac_conv2d_backward(z["x"], z["kernel"], z["cot"], k, s, p, d, dtype=np.float32)
    assert nmax(gi32, z["grad_input_f32"].astype(np.float64)) < 2e-6
    assert nmax(gk32, z["grad_kernel_f32"].astype(np.float64)) < 2e-6


@pytest.mark.parametrize("name", golden_names("g9_nd2col_"))
def test_oracle_nd2col_bit_identical(name):
    z = load_golden(name)
    g = [int(v) for v in z["geom"]]
    cols = porc.nd2col(z["x"], (g[0], g[1]), (g[2], g[3]), (g[4], g[5]), (g[8], g[9]), (g[6], g[7]), bool(g[10]))
    assert cols.shape == z["cols"].shape and np.array_equal(cols, z["cols"])


def test_oracle_fp16_case_within_half_precision():
    z = load_golden("g9_fp16")
    k, s, p, d = geom_of(z)
    out = porc.pac_conv2d_forward(z["x"], z["kernel"], k, s, p, d, dtype=np.float32)
    assert z["out"].dtype == np.float16
    assert np.abs(out - z["out"].astype(np.float32)).max() < 2e-2      # the reference multiplies and sums in half


def test_output_size_matches_oracle_over_geometries():
    rng = np.random.default_rng(5)
    n = 0
    for _ in range(400):
        H, W = (int(v) for v in rng.integers(1, 40, 2))
        k = tuple(int(v) for v in rng.integers(1, 6, 2))
        s = tuple(int(v) for v in rng.integers(1, 4, 2))
        p = tuple(int(v) for v in rng.integers(0, 5, 2))
        d = tuple(int(v) for v in rng.integers(1, 4, 2))
        want = porc.out_size((H, W), k, s, p, d)
        if min(want) < 1 or min(H + 2 * p[0] - d[0] * (k[0] - 1) - 1, W + 2 * p[1] - d[1] * (k[1] - 1) - 1) < 0:
            with pytest.raises(RuntimeError):
                pac.output_size((H, W), k, s, p, d)
            continue
        assert pac.output_size((H, W), k, s, p, d) == want
        n += 1
    assert n > 200
    assert pac.output_size((6, 7), 3, 2, 1, 1, 1, True) == porc.out_size((6, 7), 3, 2, 1, 1, 1, True) == (12, 14)


def test_host_interface_mirrors_reference():
    # argument names and order of pac.py:124, :75, :35-36
    assert list(inspect.signature(pac.conv2d).parameters) == [
        "input", "kernel", "kernel_size", "stride", "padding", "dilation", "native_impl"]
    assert list(inspect.signature(pac.Conv2dFn.forward).parameters)[1:] == [
        "input", "kernel", "kernel_size", "stride", "padding", "dilation"]
    assert list(inspect.signature(pac.nd2col).parameters) == [
        "input_nd", "kernel_size", "stride", "padding", "output_padding", "dilation", "transposed",
        "use_pyinn_if_possible"]
    with pytest.raises(ValueError, match="Incompatible input and kernel sizes"):       # pac.py:77-78 / golden manifest
        pac.conv2d(torch.zeros(1, 3, 4, 4), torch.zeros(1, 2, 3, 3, 4, 4), 3, 1, 1, 1)
    with pytest.raises(ValueError, match="does not match"):
        pac.conv2d(torch.zeros(1, 3, 4, 4), torch.zeros(1, 1, 3, 3, 5, 4), 3, 1, 1, 1)
    with pytest.raises(RuntimeError, match="ROCm device"):                              # no CPU path in the product
        pac.conv2d(torch.zeros(1, 3, 4, 4), torch.zeros(1, 1, 3, 3, 4, 4), 3, 1, 1, 1)
    with pytest.raises(RuntimeError, match="ROCm device"):
        pac.nd2col(torch.zeros(1, 1, 4, 4), 3)


@pytest.mark.parametrize("name", golden_names("g12_nd2col_grad_"))
def test_oracle_nd2col_backward_matches_reference_autograd(name):
    z = load_golden(name)
    g = [int(v) for v in z["geom"]]
    gx = porc.nd2col_backward(z["cot"].astype(np.float64), z["x"].shape[-2:], (g[0], g[1]), (g[2], g[3]), (g[4], g[5]),
                              (g[8], g[9]), (g[6], g[7]), bool(g[10]))
    assert gx.shape == z["grad_x"].shape and np.abs(gx - z["grad_x"]).max() < 1e-12
