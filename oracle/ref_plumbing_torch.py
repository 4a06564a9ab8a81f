"""PyTorch-CPU port of the *op mix* of the reference CSPN modules (cpu_baseline "port").

TEST / BENCH-BASELINE INFRASTRUCTURE ONLY (never imported by cspn_monodepth_amd/).

The reference's Python cannot travel to the GPU box, so the CPU baseline that
bench.py times next to the HIP path is this restatement of what the reference
executes per forward on a CPU: directional zero-pad of 8 gate planes, concat to
[B,8,1,H+2,W+2], and per iteration 8 pads + concat of the depth, a product, two
conv3d-with-ones channel sums, a divide, a crop and the optional sparse blend
(network/libs/post_process/CSPN_new.py:26-128); and for the K x K variant softmax,
zero centre tap, unfold * kernel summed over taps (CSPN_ours.py:24-54,
network/libs/base/pac.py:75-94).  tests/golden/make_golden.py checks it is
bit-identical to the imported reference in the build container.
"""
import torch
import torch.nn.functional as F

# (left, right, top, bottom) zero padding per gate/depth plane, CSPN_new.py:43-67
_PADS = ((0, 2, 0, 2), (1, 1, 0, 2), (2, 0, 0, 2), (0, 2, 1, 1),
         (2, 0, 1, 1), (0, 2, 2, 0), (1, 1, 2, 0), (2, 0, 2, 0))


def _eight_planes(planes):
    return torch.cat([F.pad(p, pad).unsqueeze(1) for p, pad in zip(planes, _PADS)], 1)


def cspn3_plumbing(guidance, blur_depth, sparse_depth=None, prop_time=24):
    gate_wb = _eight_planes([guidance.narrow(1, k, 1).abs() for k in range(8)])
    ones = torch.ones((1, 8, 1, 1, 1), device=guidance.device)
    mask = None if sparse_depth is None else sparse_depth.sign()
    d = blur_depth
    for _ in range(prop_time):
        d8 = _eight_planes([d] * 8)
        wsum = F.conv3d(gate_wb, ones)                 # recomputed each step, as the reference does
        tsum = F.conv3d(gate_wb * d8, ones)
        d = torch.div(tsum, wsum).squeeze(1)[:, :, 1:-1, 1:-1]
        if mask is not None:
            d = (1 - mask) * d + mask * blur_depth
    return d


def pac_plumbing(x, guided, sparse_depth=None, prop_time=24):
    B, C, H, W = guided.shape
    K = int((C + 1) ** 0.5)
    sm = F.softmax(guided, dim=1)
    kern = torch.zeros(B, C + 1, H, W, device=guided.device)
    kern[:, :C // 2] = sm[:, :C // 2]
    kern[:, C // 2 + 1:] = sm[:, C // 2:]
    kern = kern.reshape(B, 1, K, K, H, W)
    mask = None if sparse_depth is None else sparse_depth.sign()
    x0 = x
    for _ in range(prop_time):
        cols = F.unfold(x, (K, K), 1, K // 2, 1).view(B, x.shape[1], K, K, H, W)
        x = torch.einsum('ijklmn->ijmn', cols * kern).clone()
        if mask is not None:
            x = mask * x0 + (1 - mask) * x
    return x
