"""GPU parity tests of the general pixel-adaptive conv op (cspn_pac_conv2d & co through the C ABI, host mirror
cspn_monodepth_amd.base.pac) against the golden vectors captured from the reference's pac.py and against the numpy
oracle on seeded random geometries.  Tolerance: 1e-5 of the tensor's scale for fp32 (a sum of <= 49 products)."""
import numpy as np
import pytest
import torch

import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
from cspn_monodepth_amd.base import pac
from conftest import golden_names, load_golden
from oracle import cspn_oracle as orc
from oracle import pac_oracle as porc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5
CONV_CASES = [n for n in golden_names("g9_") if "nd2col" not in n and n != "g9_fp16"]


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t if dtype is None else t.to(dtype)


class force_generic(object):
    """with force_generic(1): every cspn_pac_* call runs the generic kernels (include/cspn_hip.h cspn_pac_force_generic)."""

    def __init__(self, on):
        self.on = int(on)

    def __enter__(self):
        import ctypes
        from cspn_monodepth_amd import _lib
        self.prev = ctypes.c_int(0)
        assert _lib.lib().cspn_pac_force_generic(self.on, ctypes.byref(self.prev))

    def __exit__(self, *exc):
        from cspn_monodepth_amd import _lib
        assert _lib.lib().cspn_pac_force_generic(self.prev.value, None)
        return False


def geom_of(z):
    kh, kw, sh, sw, ph, pw, dh, dw = (int(v) for v in z["geom"][:8])
    return (kh, kw), (sh, sw), (ph, pw), (dh, dw)


def nmax(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def run_all(x, kern, cot, k, s, p, d, dtype=torch.float32):
    xt = dev(x, dtype).requires_grad_(True)
    kt = dev(kern, dtype).requires_grad_(True)
    out = pac.conv2d(xt, kt, k, s, p, d)
    out.backward(dev(cot, dtype))
    torch.cuda.synchronize()
    return (out.detach().float().cpu().numpy(), xt.grad.float().cpu().numpy(), kt.grad.float().cpu().numpy())


@pytest.mark.parametrize("name", CONV_CASES)
def test_golden_forward_and_backward(name):
    z = load_golden(name)
    k, s, p, d = geom_of(z)
    out, gi, gk = run_all(z["x"], z["kernel"], z["cot"], k, s, p, d)
    assert out.dtype == np.float32
    assert nmax(out, z["out"]) <= TOL, name
    assert nmax(gi, z["grad_input_f64"]) <= TOL, name
    assert nmax(gk, z["grad_kernel_f64"]) <= TOL, name
    with torch.no_grad():                                   # no-grad route and native_impl flag: same kernel
        o2 = pac.conv2d(dev(z["x"]), dev(z["kernel"]), k, s, p, d, native_impl=True)
    assert np.array_equal(o2.cpu().numpy(), out)


@pytest.mark.parametrize("name", golden_names("g9_nd2col_"))
def test_golden_nd2col_bit_exact(name):
    z = load_golden(name)
    g = [int(v) for v in z["geom"]]
    cols = pac.nd2col(dev(z["x"]), (g[0], g[1]), (g[2], g[3]), (g[4], g[5]), (g[8], g[9]), (g[6], g[7]), bool(g[10]))
    assert tuple(cols.shape) == z["cols"].shape
    assert np.array_equal(cols.cpu().numpy(), z["cols"])
    c16 = pac.nd2col(dev(z["x"], torch.float16), (g[0], g[1]), (g[2], g[3]), (g[4], g[5]), (g[8], g[9]), (g[6], g[7]),
                     bool(g[10]))
    assert np.array_equal(c16.cpu().numpy(), z["cols"].astype(np.float16))


def test_golden_fp16():
    z = load_golden("g9_fp16")
    k, s, p, d = geom_of(z)
    with torch.no_grad():
        out = pac.conv2d(dev(z["x"]), dev(z["kernel"]), k, s, p, d)
    assert out.dtype == torch.float16
    exact = porc.pac_conv2d_forward(z["x"], z["kernel"], k, s, p, d, dtype=np.float32)
    got = out.float().cpu().numpy()
    # fp32 accumulation, one rounding: within half an fp16 ulp of the exact sum, and at least as close as the reference
    assert np.abs(got - exact).max() <= np.abs(exact).max() * 2.0 ** -11 + 1e-7
    assert np.abs(got - exact).max() <= np.abs(z["out"].astype(np.float32) - exact).max() + 1e-7


def _random_geometry(rng):
    while True:
        H, W = (int(v) for v in rng.integers(1, 48, 2))
        k = tuple(int(v) for v in rng.integers(1, 6, 2))
        s = tuple(int(v) for v in rng.integers(1, 4, 2))
        p = tuple(int(v) for v in rng.integers(0, 5, 2))
        d = tuple(int(v) for v in rng.integers(1, 4, 2))
        if min(H + 2 * p[0] - d[0] * (k[0] - 1) - 1, W + 2 * p[1] - d[1] * (k[1] - 1) - 1) >= 0:
            return H, W, k, s, p, d


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_geometry_vs_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    H, W, k, s, p, d = _random_geometry(rng)
    B, C = int(rng.integers(1, 4)), int(rng.integers(1, 10))
    CK = C if rng.integers(0, 2) else 1
    Ho, Wo = porc.out_size((H, W), k, s, p, d)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    kern = rng.standard_normal((B, CK, k[0], k[1], Ho, Wo)).astype(np.float32)
    cot = rng.standard_normal((B, C, Ho, Wo)).astype(np.float32)
    out, gi, gk = run_all(x, kern, cot, k, s, p, d)
    want = porc.pac_conv2d_forward(x, kern, k, s, p, d, dtype=np.float64)
    wgi, wgk = porc.pac_conv2d_backward(x, kern, cot, k, s, p, d)
    tag = (B, C, CK, H, W, k, s, p, d)
    assert nmax(out, want) <= TOL, tag
    assert nmax(gi, wgi) <= TOL, tag
    assert nmax(gk, wgk) <= TOL, tag
    cols = pac.nd2col(dev(x), k, s, p, 0, d).cpu().numpy()
    assert np.array_equal(cols, porc.nd2col(x, k, s, p, 0, d)), tag


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_tiled_geometry_both_kernels(seed):
    """Stride 1 / dilation 1 / square K in {3,5,7} takes the LDS-tiled kernels; cspn_pac_force_generic(1) forces the generic
    ones.  Both must match the oracle on frames that span several 64 x 16 tiles, ragged edges included."""
    rng = np.random.default_rng(2000 + seed)
    K = (3, 5, 7)[seed % 3]
    p = (int(rng.integers(0, K + 1)), int(rng.integers(0, K + 1)))
    H, W = int(rng.integers(max(1, K - 2 * p[0]), 60)), int(rng.integers(max(1, K - 2 * p[1]), 200))
    if seed % 4 == 0:
        W = (W + 3) // 4 * 4 - 2 * p[1] + K - 1            # Wo % 4 == 0: the vector path
        W = max(W, 4)
    B, C = int(rng.integers(1, 3)), int(rng.integers(1, 11))
    CK = C if seed % 2 else 1
    dt = torch.float16 if seed % 6 == 5 else torch.float32
    Ho, Wo = porc.out_size((H, W), K, 1, p, 1)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    kern = (rng.standard_normal((B, CK, K, K, Ho, Wo)) * 0.3).astype(np.float32)
    cot = rng.standard_normal((B, C, Ho, Wo)).astype(np.float32)
    if dt == torch.float16:
        x, kern, cot = (v.astype(np.float16).astype(np.float32) for v in (x, kern, cot))
    want = porc.pac_conv2d_forward(x, kern, K, 1, p, 1, dtype=np.float64)
    wgi, wgk = porc.pac_conv2d_backward(x, kern, cot, K, 1, p, 1)
    tol = 2e-3 if dt == torch.float16 else TOL
    tag = (B, C, CK, H, W, K, p, str(dt))
    for scalar in (0, 1):
        with force_generic(scalar):
            out, gi, gk = run_all(x, kern, cot, K, 1, p, 1, dt)
        assert nmax(out, want) <= tol, (tag, scalar)
        assert nmax(gi, wgi) <= tol, (tag, scalar)
        assert nmax(gk, wgk) <= tol, (tag, scalar)


@pytest.mark.parametrize("seed", range(32))
def test_fuzz_stride2_register_kernels_and_generic(seed):
    """Stride 2, dilation 1, K in {3, 5}, padding K / 2, W % 8 == 0 takes the register / DPP kernels of pac_conv2d_s2.hip (forward and
    both gradients); cspn_pac_force_generic(1) forces the generic ones.  Both against the oracle: odd and even H (the last input
    row pair is half empty), H = 1, one octet per row, rows wider than a wavefront (the first / last lane of a wavefront patches
    its fringe with scalar loads), ragged channel batches, shared and per-channel kernels, fp16."""
    rng = np.random.default_rng(5000 + seed)
    K = (3, 5)[seed % 2]
    P = K // 2
    H = int(rng.integers(1, 40)) if seed % 8 else (1, 2, 3, 5)[(seed // 8) % 4]
    W = 8 * int(rng.integers(1, 12))
    if seed % 5 == 0:
        W = 8 * int(rng.integers(65, 160))                  # more than 64 octets per row: wave boundaries inside a row
        H = int(rng.integers(1, 9))
    B, C = int(rng.integers(1, 3)), int(rng.integers(1, 10))
    CK = C if (seed // 2) % 2 else 1
    dt = torch.float16 if seed % 4 == 3 else torch.float32
    Ho, Wo = porc.out_size((H, W), K, 2, P, 1)
    assert Wo == W // 2 and Ho == (H + 1) // 2
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    kern = (rng.standard_normal((B, CK, K, K, Ho, Wo)) * 0.3).astype(np.float32)
    cot = rng.standard_normal((B, C, Ho, Wo)).astype(np.float32)
    if dt == torch.float16:
        x, kern, cot = (v.astype(np.float16).astype(np.float32) for v in (x, kern, cot))
    want = porc.pac_conv2d_forward(x, kern, K, 2, P, 1, dtype=np.float64)
    wgi, wgk = porc.pac_conv2d_backward(x, kern, cot, K, 2, P, 1)
    tol = 2e-3 if dt == torch.float16 else TOL
    tag = (B, C, CK, H, W, K, str(dt))
    for scalar in (0, 1):
        with force_generic(scalar):
            out, gi, gk = run_all(x, kern, cot, K, 2, P, 1, dt)
        assert nmax(out, want) <= tol, (tag, scalar)
        assert nmax(gi, wgi) <= tol, (tag, scalar)
        assert nmax(gk, wgk) <= tol, (tag, scalar)


def test_stride2_unaligned_tensors_take_the_generic_kernels():
    """The register kernels need 16-byte aligned bases; a tensor that starts 4 bytes into its storage must still give the right
    answer (the entry points fall back to the generic kernels instead of issuing misaligned 16-byte accesses)."""
    torch.manual_seed(5)
    B, C, H, W, K = 2, 3, 10, 16, 3
    Ho, Wo = pac.output_size((H, W), K, 2, 1, 1)

    def off1(shape):
        n = int(np.prod(shape))
        return torch.randn(n + 1, device=DEV)[1:].view(shape)
    x, k, g = off1((B, C, H, W)), off1((B, 1, K, K, Ho, Wo)).mul_(0.3), off1((B, C, Ho, Wo))
    assert x.data_ptr() % 16 and k.data_ptr() % 16 and g.data_ptr() % 16 and x.is_contiguous()
    xa, ka = x.clone().requires_grad_(True), k.clone().requires_grad_(True)        # aligned copies: the register kernels
    xu, ku = x.requires_grad_(True), k.requires_grad_(True)
    oa = pac.conv2d(xa, ka, K, 2, 1, 1); oa.backward(g.clone())
    ou = pac.conv2d(xu, ku, K, 2, 1, 1); ou.backward(g)
    for a, b in ((oa, ou), (xa.grad, xu.grad), (ka.grad, ku.grad)):
        assert float((a - b).abs().max() / a.abs().max()) <= 2e-6


@pytest.mark.parametrize("C,CK,K,dt", [(32, 1, 3, torch.float32), (9, 9, 3, torch.float32), (16, 1, 5, torch.float32),
                                       (6, 6, 5, torch.float32), (12, 1, 3, torch.float16)])
def test_stride2_full_frame_register_kernels_equal_generic(C, CK, K, dt):
    """NYU- and KITTI-size frames at stride 2 (many workgroups, channel chunks over blockIdx.y, the wavefront-split channel sum of
    the shared dL/dkernel): the register kernels against the generic ones on the same tensors."""
    for B, H, W in ((2, 228, 304), (1, 127, 1216)):
        torch.manual_seed(21)
        P = K // 2
        Ho, Wo = pac.output_size((H, W), K, 2, P, 1)
        x = torch.randn(B, C, H, W, device=DEV).to(dt).requires_grad_(True)
        k = (torch.randn(B, CK, K, K, Ho, Wo, device=DEV) * 0.3).to(dt).requires_grad_(True)
        g = torch.randn(B, C, Ho, Wo, device=DEV).to(dt)
        res = []
        for scalar in (0, 1):
            x.grad = k.grad = None
            with force_generic(scalar):
                out = pac.conv2d(x, k, K, 2, P, 1)
                out.backward(g)
            res.append((out.detach().float(), x.grad.float().clone(), k.grad.float().clone()))
        tol = 2e-3 if dt == torch.float16 else 2e-6
        for got, want in zip(*res):
            err = float((got - want).abs().max() / want.abs().max())
            assert err <= tol, (C, CK, K, str(dt), H, W, err)


@pytest.mark.parametrize("seed", range(36))
def test_fuzz_unit_stride_dilated_and_rectangular_windows(seed):
    """Unit stride with dilation / non-square / even windows takes the any-geometry LDS-tiled kernels — forward, dL/dinput
    (the same kernel transposed) and dL/dkernel (tap groups of a shared kernel summed over channels in registers, outer
    product for a per-channel kernel); compare them and the scalar kernels (cspn_pac_force_generic) with the oracle on
    multi-tile frames, ragged edges, more channels than one LDS batch, fp16 now and then."""
    rng = np.random.default_rng(3000 + seed)
    k = (int(rng.integers(1, 6)), int(rng.integers(1, 6)))
    d = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
    if k[0] == k[1] and k[0] in (3, 5, 7) and d == (1, 1):
        d = (2, 1)
    p = (int(rng.integers(0, 6)), int(rng.integers(0, 6)))
    H = int(rng.integers(max(1, d[0] * (k[0] - 1) + 1 - 2 * p[0]), 50))
    W = int(rng.integers(max(1, d[1] * (k[1] - 1) + 1 - 2 * p[1]), 180))
    B, C = int(rng.integers(1, 3)), int(rng.integers(1, 10))
    CK = C if seed % 2 else 1
    dt = torch.float16 if seed % 6 == 4 else torch.float32
    Ho, Wo = porc.out_size((H, W), k, 1, p, d)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    kern = (rng.standard_normal((B, CK, k[0], k[1], Ho, Wo)) * (0.3 if dt == torch.float16 else 1.0)).astype(np.float32)
    cot = rng.standard_normal((B, C, Ho, Wo)).astype(np.float32)
    if dt == torch.float16:
        x, kern, cot = (v.astype(np.float16).astype(np.float32) for v in (x, kern, cot))
    want = porc.pac_conv2d_forward(x, kern, k, 1, p, d, dtype=np.float64)
    wgi, wgk = porc.pac_conv2d_backward(x, kern, cot, k, 1, p, d)
    tol = 2e-3 if dt == torch.float16 else TOL
    for scalar in (0, 1):
        tag = (B, C, CK, H, W, k, p, d, str(dt), scalar)
        with force_generic(scalar):
            out, gi, gk = run_all(x, kern, cot, k, 1, p, d, dt)
        assert nmax(out, want) <= tol, tag
        assert nmax(gi, wgi) <= tol, tag
        assert nmax(gk, wgk) <= tol, tag


@pytest.mark.parametrize("C,CK,k,p,d", [(32, 1, (3, 3), (2, 2), (2, 2)), (6, 6, (3, 3), (2, 2), (2, 2)), (5, 1, (5, 5), (4, 4), (2, 2)),
                                         (3, 1, (2, 4), (1, 3), (3, 2)), (4, 4, (4, 1), (0, 0), (1, 1))])
def test_dilated_windows_full_frame_tiled_equals_generic(C, CK, k, p, d):
    """At frame size (228 x 304: 5 x 15 tiles, every tap-group split of a shared kernel) the tiled any-geometry kernels
    against the scalar ones."""
    rng = np.random.default_rng(77)
    B, H, W = 2, 228, 304
    Ho, Wo = porc.out_size((H, W), k, 1, p, d)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    kern = rng.standard_normal((B, CK, k[0], k[1], Ho, Wo)).astype(np.float32)
    cot = rng.standard_normal((B, C, Ho, Wo)).astype(np.float32)
    with force_generic(0):
        got = run_all(x, kern, cot, k, 1, p, d)
    with force_generic(1):
        ref = run_all(x, kern, cot, k, 1, p, d)
    for g_, r_, what in zip(got, ref, ("out", "grad_input", "grad_kernel")):
        assert nmax(g_, r_) <= 2e-6, (what, C, CK, k, p, d)


@pytest.mark.parametrize("B,C,H,W", [(4, 8, 228, 304), (4, 10, 228, 304), (16, 12, 128, 512), (4, 3, 228, 304)])
def test_shared_7x7_with_many_tiles_tiled_equals_generic(B, C, H, W):
    """A shared 7 x 7 kernel on launches with >= 256 tiles takes the eight-channel batches (one LDS buffer when the chunk is a
    single batch, two when it is not: the 1024-tile case) and the whole-window dL/dkernel; fewer than eight channels keep the
    four-channel kernel.  All of them against the scalar kernels."""
    rng = np.random.default_rng(91)
    K, p = 7, 3
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    kern = (rng.standard_normal((B, 1, K, K, H, W)) * 0.2).astype(np.float32)
    cot = rng.standard_normal((B, C, H, W)).astype(np.float32)
    with force_generic(0):
        got = run_all(x, kern, cot, K, 1, p, 1)
    with force_generic(1):
        ref = run_all(x, kern, cot, K, 1, p, 1)
    for g_, r_, what in zip(got, ref, ("out", "grad_input", "grad_kernel")):
        assert nmax(g_, r_) <= 2e-6, (what, B, C, H, W)


def test_padding_zero_times_nonfinite_kernel_is_nan_like_unfold():
    # F.unfold gives 0 in the padding and the reference multiplies it by the kernel: 0 * inf = NaN (pac.py:89-92)
    x = np.ones((1, 1, 4, 4), np.float32)
    kern = np.ones((1, 1, 3, 3, 4, 4), np.float32)
    kern[0, 0, 0, 0, 0, 0] = np.inf            # tap (-1,-1) at the corner pixel looks at the padding
    kern[0, 0, 1, 1, 2, 2] = np.inf            # centre tap at an interior pixel: a real inf
    with torch.no_grad():
        out = pac.conv2d(dev(x), dev(kern), 3, 1, 1, 1).cpu().numpy()
    with np.errstate(invalid="ignore"):
        want = porc.pac_conv2d_forward(x, kern, 3, 1, 1, 1)
    assert np.isnan(out[0, 0, 0, 0]) and np.isposinf(out[0, 0, 2, 2])
    assert np.array_equal(np.isnan(out), np.isnan(want)) and np.array_equal(np.isinf(out), np.isinf(want))


@pytest.mark.parametrize("K", [3, 5, 7])
def test_one_step_of_the_recurrence_equals_the_general_op(K):
    """Cross-check of the two HIP paths at full frame size: CSPN_ours' step (CSPN_ours.py:47-53) is
    conv2d(x, kernel, K, 1, K//2, 1) with the softmax kernel and a zero centre tap."""
    B, H, W = 4, 228, 304
    C = K * K - 1
    gd = dev(orc.hash_normal(70 + K, 1, (B, C, H, W)))
    x = dev(orc.hash_uniform(70 + K, 2, (B, 1, H, W), 0.0, 10.0))
    with torch.no_grad():
        step = pkg.CSPN_ours.AffinityPropagate(1)(x, gd)
        sm = torch.softmax(gd, dim=1)
        kern = torch.zeros(B, C + 1, H, W, device=DEV)
        kern[:, :C // 2] = sm[:, :C // 2]
        kern[:, C // 2 + 1:] = sm[:, C // 2:]
        gen = pac.conv2d(x, kern.view(B, 1, K, K, H, W), K, 1, K // 2, 1)
    assert nmax(gen.cpu().numpy(), step.cpu().numpy()) <= 2e-6


def test_linearity_and_adjointness_at_full_size():
    """Size-independent properties at the NYU frame: the op is bilinear, and <conv(x,k), g> = <x, grad_input(g,k)>
    = <k, grad_kernel(g,x)> (the gradients are the exact adjoints)."""
    B, C, H, W, K = 2, 3, 228, 304, 5
    torch.manual_seed(3)
    x = torch.randn(B, C, H, W, device=DEV, dtype=torch.float32)
    k = torch.randn(B, 1, K, K, H, W, device=DEV, dtype=torch.float32)
    g = torch.randn(B, C, H, W, device=DEV, dtype=torch.float32)
    xr, kr = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
    out = pac.conv2d(xr, kr, K, 1, K // 2, 1)
    out.backward(g)
    lhs = float((out.detach().double() * g.double()).sum())
    assert abs(lhs - float((x.double() * xr.grad.double()).sum())) <= 1e-5 * abs(lhs) + 1e-3
    assert abs(lhs - float((k.double() * kr.grad.double()).sum())) <= 1e-5 * abs(lhs) + 1e-3
    with torch.no_grad():
        o2 = pac.conv2d(2.0 * x, k, K, 1, K // 2, 1)
    assert torch.equal(o2, 2.0 * out.detach())              # scaling by 2 is exact in binary floating point


@pytest.mark.parametrize("C,CK,K,s,p,d", [(16, 1, 5, 1, 2, 1), (8, 8, 3, 1, 1, 1), (6, 1, 7, 1, 3, 1), (8, 1, 3, 2, 1, 1), (5, 5, 5, 2, 2, 1),
                                          (8, 1, 3, 1, 2, 2), (5, 5, (3, 2), 1, (1, 0), (1, 3))])
def test_full_frame_against_the_unfold_formulation(C, CK, K, s, p, d):
    """NYU-size frames, many tiles and channel batches: forward and both gradients against the reference's own
    formulation (F.unfold * kernel, summed — pac.py:89-92 / :130-140) evaluated with stock torch ops and autograd on the
    same GPU in fp64.  An independent implementation, not the oracle: it checks scale, tiling seams and channel batching."""
    B, H, W = 2, 228, 304
    torch.manual_seed(11)
    kh, kw = (K, K) if isinstance(K, int) else K
    Ho, Wo = pac.output_size((H, W), K, s, p, d)
    x = torch.randn(B, C, H, W, device=DEV)
    k = torch.randn(B, CK, kh, kw, Ho, Wo, device=DEV) * 0.3
    g = torch.randn(B, C, Ho, Wo, device=DEV)
    xr, kr = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
    out = pac.conv2d(xr, kr, K, s, p, d)
    out.backward(g)
    x64, k64 = x.double().requires_grad_(True), k.double().requires_grad_(True)
    cols = torch.nn.functional.unfold(x64, (kh, kw), d, p, s).view(B, C, kh, kw, Ho, Wo)
    ref = (cols * k64).sum(dim=(2, 3))
    ref.backward(g.double())
    for got, want in ((out, ref), (xr.grad, x64.grad), (kr.grad, k64.grad)):
        err = float((got.detach().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert err <= TOL, (C, CK, K, s, p, d, err)


def test_only_needed_gradients_are_computed():
    x = torch.randn(1, 2, 8, 8, device=DEV)
    k = torch.randn(1, 1, 3, 3, 8, 8, device=DEV, requires_grad=True)
    pac.conv2d(x, k, 3, 1, 1, 1).sum().backward()
    assert k.grad is not None and x.grad is None
    x2 = x.clone().requires_grad_(True)
    pac.conv2d(x2, k.detach(), 3, 1, 1, 1).sum().backward()
    assert x2.grad is not None


@pytest.mark.parametrize("name", golden_names("g12_nd2col_grad_"))
def test_nd2col_is_differentiable(name):
    """the reference's nd2col is differentiable (pac.py:51-68 via F.unfold); the HIP nd2col's backward
    is the fold.  Golden = reference autograd of sum(cols * cot)."""
    z = load_golden(name)
    g = [int(v) for v in z["geom"]]
    x = dev(z["x"]).requires_grad_(True)
    cols = pac.nd2col(x, (g[0], g[1]), (g[2], g[3]), (g[4], g[5]), (g[8], g[9]), (g[6], g[7]), bool(g[10]))
    assert cols.requires_grad
    (cols * dev(z["cot"])).sum().backward()
    want = z["grad_x"]
    assert x.grad.shape == want.shape
    assert float(np.abs(x.grad.cpu().numpy() - want).max()) <= 2e-6 * max(1.0, float(np.abs(want).max()))
    # no-grad call: plain tensor, same values
    with torch.no_grad():
        c2 = pac.nd2col(x, (g[0], g[1]), (g[2], g[3]), (g[4], g[5]), (g[8], g[9]), (g[6], g[7]), bool(g[10]))
    assert not c2.requires_grad and torch.equal(c2, cols.detach())


def test_native_impl_formulation_backpropagates_through_nd2col():
    """pac.py:130-140 written out with this package's nd2col: gradients equal Conv2dFn's (G9 golden)."""
    z = load_golden("g9_k3_same_shared")
    k, s, p, d = geom_of(z)
    x, kern = dev(z["x"]).requires_grad_(True), dev(z["kernel"]).requires_grad_(True)
    cols = pac.nd2col(x, k, stride=s, padding=p, dilation=d)
    out = (cols * kern).sum(dim=(2, 3))
    out.backward(dev(z["cot"]))
    assert float(np.abs(out.detach().cpu().numpy() - z["out"]).max()) < 1e-5
    for got, key in ((x.grad, "grad_input_f64"), (kern.grad, "grad_kernel_f64")):
        want = z[key]
        assert float(np.abs(got.cpu().numpy() - want).max()) <= 2e-6 * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("K", [3, 5])
@pytest.mark.parametrize("C,CK", [(1, 1), (3, 1), (3, 3)])
@pytest.mark.parametrize("Ho,Wo,same", [(19, 136, True), (33, 304, True), (5, 8, True), (21, 264, False)])
def test_fp16_eight_pixel_kernels(K, C, CK, Ho, Wo, same):
    """fp16 with whole 16-byte octs (Wo % 8 == 0) takes pac_conv2d_tiled_h8 for the forward and, without a channel sum
    (CK == C), pac_conv2d_gk_h8 for dL/dkernel; the input gradient loads its guarded halfs raw and converts late.  Several
    128 x 16 tiles with ragged last rows / columns, 'same' and no padding, against the oracle on the fp16-rounded inputs and
    against the generic kernels."""
    rng = np.random.default_rng(7000 + 100 * K + 10 * C + CK + Ho + Wo)
    p = (K // 2, K // 2) if same else (0, 0)
    H, W = Ho + K - 1 - 2 * p[0], Wo + K - 1 - 2 * p[1]
    B = 2
    x = rng.standard_normal((B, C, H, W)).astype(np.float16).astype(np.float32)
    kern = (rng.standard_normal((B, CK, K, K, Ho, Wo)) * 0.3).astype(np.float16).astype(np.float32)
    cot = rng.standard_normal((B, C, Ho, Wo)).astype(np.float16).astype(np.float32)
    assert porc.out_size((H, W), K, 1, p, 1) == (Ho, Wo)
    want = porc.pac_conv2d_forward(x, kern, K, 1, p, 1, dtype=np.float64)
    wgi, wgk = porc.pac_conv2d_backward(x, kern, cot, K, 1, p, 1)
    res = {}
    for scalar in (0, 1):
        with force_generic(scalar):
            res[scalar] = run_all(x, kern, cot, K, 1, p, 1, torch.float16)
        out, gi, gk = res[scalar]
        assert nmax(out, want) <= 2e-3 and nmax(gi, wgi) <= 2e-3 and nmax(gk, wgk) <= 2e-3, (scalar, nmax(out, want), nmax(gi, wgi), nmax(gk, wgk))
    # same fp32 accumulation order per output in both forward kernels and an exact product in dL/dkernel: identical halfs
    assert np.array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize("K", [3, 5])
def test_fp16_eight_pixel_kernel_several_channels_per_workgroup(K):
    """pac_conv2d_tiled_h8 with a per-channel kernel (CK == C) and MORE THAN ONE channel per workgroup (cchunk > 1: the
    launcher stops splitting channels once tiles x batch x chunks >= 4096).  The tap rows are double-buffered across the
    channel boundary — row 0 of the next channel is prefetched while the last row of the current one is applied — which
    round 2 got wrong for every channel after a workgroup's first.  Too large for the numpy oracle (the kernel
    tensor is ~0.6 GB): the tiled kernel must equal the generic one, which accumulates in the same order, half for half,
    and a few outputs are checked against a direct fp64 evaluation."""
    B, C, Ho, Wo = 64, 5, 64, 512                      # 16 tiles x 64 images -> 4 chunks of (2, 2, 1) channels
    g = torch.Generator(device="cuda").manual_seed(900 + K)
    x = torch.randn(B, C, Ho, Wo, device="cuda", generator=g).half()
    kern = (torch.randn(B, C, K, K, Ho, Wo, device="cuda", generator=g) * 0.3).half()
    outs = {}
    with torch.no_grad():
        for scalar in (0, 1):
            with force_generic(scalar):
                outs[scalar] = pac.conv2d(x, kern, K, 1, K // 2, 1)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    xp = torch.nn.functional.pad(x.double(), (K // 2,) * 4)
    rng = np.random.default_rng(5)
    for _ in range(64):
        b, c, y, xx = (int(rng.integers(0, n)) for n in (B, C, Ho, Wo))
        want = float((xp[b, c, y:y + K, xx:xx + K] * kern[b, c, :, :, y, xx].double()).sum())
        assert abs(float(outs[0][b, c, y, xx]) - want) <= 2e-3 * max(1.0, abs(want)), (b, c, y, xx)
