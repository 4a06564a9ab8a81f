"""Zero-insertion un-pooling: oracle vs the reference's goldens (CPU), HIP kernels vs goldens and
oracle, bit-exact (GPU).  Fixtures: tests/golden/g10_unpool_*.npz from make_golden_unpool.py."""
import inspect

import numpy as np
import pytest
import torch

import cspn_monodepth_amd as pkg
from cspn_monodepth_amd.network import up_pooling as up
from conftest import golden_names, load_golden
from oracle import pac_oracle as porc

CASES = golden_names("g10_unpool_")
DEV = "cuda:0"


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    z = load_golden(name)
    s, oh, ow = (int(v) for v in z["geom"])
    assert np.array_equal(porc.up_pooling(z["x"], s, oh, ow), z["out"])
    assert np.array_equal(porc.up_pooling_backward(z["cot"], z["x"].shape[-2:], s), z["grad_x"])


def test_host_mirror_contract():
    b = up.MyBlock(10, 12)                                   # unet_ours.py:133-136
    assert (b.oheight, b.owidth) == (10, 12) and len(b.state_dict()) == 0
    assert list(inspect.signature(b._up_pooling).parameters) == ["x", "scale"]
    with pytest.raises(ValueError, match="not supported"):   # the reference's 0 defaults: empty / all-zero output
        up.MyBlock()._up_pooling(torch.zeros(1, 1, 2, 2), 2)
    with pytest.raises(ValueError):
        up.up_pooling(torch.zeros(1, 1, 2, 2), 2, 5, 4)      # larger than scale * H
    with pytest.raises(RuntimeError, match="ROCm device"):   # no CPU path
        up.up_pooling(torch.zeros(1, 1, 2, 2), 2, 4, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_hip_matches_golden_bit_exact(name, dtype):
    z = load_golden(name)
    s, oh, ow = (int(v) for v in z["geom"])
    x = torch.from_numpy(z["x"]).to(DEV, dtype).requires_grad_(True)
    y = up.MyBlock(oh, ow)._up_pooling(x, s)
    y.backward(torch.from_numpy(z["cot"]).to(DEV, dtype))
    ref_dt = np.float16 if dtype == torch.float16 else np.float32
    assert y.dtype == dtype and tuple(y.shape) == z["out"].shape
    assert np.array_equal(y.detach().cpu().numpy(), porc.up_pooling(z["x"].astype(ref_dt), s, oh, ow))
    assert np.array_equal(x.grad.cpu().numpy(), porc.up_pooling_backward(z["cot"].astype(ref_dt), z["x"].shape[-2:], s))
    if dtype == torch.float32:
        assert np.array_equal(y.detach().cpu().numpy(), z["out"]) and np.array_equal(x.grad.cpu().numpy(), z["grad_x"])


@pytest.mark.gpu
def test_hip_decoder_sizes_and_garbage_free_output():
    # the five decoder stages of unet_cspn_nyu (228 x 304 input): every output element is written, zeros included
    for (H, W, oh, ow, C) in ((8, 10, 15, 19, 64), (15, 19, 29, 38, 32), (29, 38, 57, 76, 16), (57, 76, 114, 152, 8),
                              (114, 152, 228, 304, 4)):
        x = torch.randn(2, C, H, W, device=DEV)
        torch.empty(2, C, oh, ow, device=DEV).fill_(float("nan"))          # poison the allocator's next block
        with torch.no_grad():
            y = up.up_pooling(x, 2, oh, ow)
        want = torch.zeros(2, C, 2 * H, 2 * W, device=DEV)
        want[:, :, ::2, ::2] = x
        assert torch.equal(y, want[:, :, :oh, :ow])
    # non-finite activations: both reference formulations MULTIPLY by the zero weight / mask, so the whole 2 x 2 block
    # of a NaN or inf input becomes NaN (captured from the reference: see the expected tensor below)
    x = torch.tensor([[[[float("nan"), 1.0], [2.0, float("inf")]]]], device=DEV)
    y = up.up_pooling(x, 2, 4, 4).cpu().numpy()[0, 0]
    n = float("nan")
    want = np.array([[n, n, 1, 0], [n, n, 0, 0], [2, 0, np.inf, n], [0, 0, n, n]], np.float32)
    assert np.array_equal(y, want, equal_nan=True)
