#!/usr/bin/env python3
"""Roofline of the general pixel-adaptive conv op (cspn_pac_conv2d & gradients) on the GPU box (developer tool).

    python tools/bench_pac_conv.py [--json gpurun_out/pac_conv.json]

Algorithmic bytes per output pixel: kernel_ch*kh*kw (kernel) + C (input, read once) + C (output) elements for the
forward; the gradients move the same tensors the other way.  Also times the reference's formulation (F.unfold *
kernel, summed — pac.py:89-92) with torch ops on the same GPU for scale.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cspn_monodepth_amd.base import pac            # noqa: E402
from tools.tune import timed as _timed_once        # noqa: E402


def timed(fn, reps):
    """Best of three batches: one allocator event (a cached block split or returned between cases) inside a batch of 20
    otherwise shows up as milliseconds on a 15 us kernel."""
    return min(_timed_once(fn, reps) for _ in range(3))

PEAK = 8000.0
CASES = [   # name, B, C, CK, H, W, K, stride, pad, dil, dtype
    ("c1_k5_same (CSPN_ours step)", 24, 1, 1, 228, 304, 5, 1, 2, 1, torch.float32),
    ("c1_k3_same", 24, 1, 1, 228, 304, 3, 1, 1, 1, torch.float32),
    ("c1_k5_same_f16", 24, 1, 1, 228, 304, 5, 1, 2, 1, torch.float16),
    ("c64_k5_shared", 8, 64, 1, 228, 304, 5, 1, 2, 1, torch.float32),
    ("c64_k3_perch", 8, 64, 64, 228, 304, 3, 1, 1, 1, torch.float32),
    ("c32_k3_stride2", 8, 32, 1, 228, 304, 3, 2, 1, 1, torch.float32),
    ("c32_k5_stride2", 8, 32, 1, 228, 304, 5, 2, 2, 1, torch.float32),
    ("c32_k3_stride2_perch", 8, 32, 32, 228, 304, 3, 2, 1, 1, torch.float32),
    ("c32_k3_stride2_f16", 8, 32, 1, 228, 304, 3, 2, 1, 1, torch.float16),
    ("c32_k3_dil2", 8, 32, 1, 228, 304, 3, 1, 2, 2, torch.float32),
    ("c16_k7_shared", 8, 16, 1, 228, 304, 7, 1, 3, 1, torch.float32),
]


def unfold_formulation(x, k, K, s, p, d):
    B, C = x.shape[:2]
    cols = torch.nn.functional.unfold(x, K, d, p, s).view(B, C, K, K, *k.shape[-2:])
    return (cols * k).sum(dim=(2, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", help="substring filter on the case names (skips the un-pooling rows)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rows = []
    for name, B, C, CK, H, W, K, s, p, d, dt in CASES:
        if args.only and args.only not in name:
            continue
        Ho, Wo = pac.output_size((H, W), K, s, p, d)
        x = torch.randn(B, C, H, W, device=dev, dtype=dt)
        k = torch.randn(B, CK, K, K, Ho, Wo, device=dev, dtype=dt)
        g = torch.randn(B, C, Ho, Wo, device=dev, dtype=dt)
        es = x.element_size()
        fwd_bytes = (k.numel() + x.numel() + g.numel()) * es
        with torch.no_grad():
            t_f = timed(lambda: pac.conv2d(x, k, K, s, p, d), args.reps)
            t_u = timed(lambda: unfold_formulation(x, k, K, s, p, d), 5)
        geom = pac._geometry(K, s, p, d)                 # the two gradient kernels alone, without autograd's host time
        t_gi = timed(lambda: pac._grad_input(g, k, tuple(x.shape), geom), args.reps)
        t_gk = timed(lambda: pac._grad_kernel(g, x, CK, geom), args.reps)
        t_b = t_gi + t_gk
        bwd_bytes = (2 * k.numel() + 2 * x.numel() + 2 * g.numel()) * es     # gi: g,k -> gx ; gk: g,x -> gk
        row = dict(case=name, B=B, C=C, kernel_ch=CK, H=H, W=W, K=K, stride=s, padding=p, dilation=d,
                   dtype=str(dt).split(".")[-1], fwd_us=t_f, fwd_GBs=fwd_bytes / t_f / 1e3, fwd_frac=fwd_bytes / t_f / 1e3 / PEAK,
                   bwd_us=t_b, bwd_GBs=bwd_bytes / t_b / 1e3, bwd_frac=bwd_bytes / t_b / 1e3 / PEAK,
                   grad_input_us=t_gi, grad_kernel_us=t_gk, torch_unfold_us=t_u)
        rows.append(row)
        print("%-30s fwd %8.1f us %6.0f GB/s (%4.1f%%)   bwd %8.1f us %6.0f GB/s (%4.1f%%) [gi %.1f gk %.1f]   torch unfold fwd %9.1f us (x%.1f)" % (
            name, t_f, row["fwd_GBs"], 100 * row["fwd_frac"], t_b, row["bwd_GBs"], 100 * row["bwd_frac"], t_gi, t_gk, t_u, t_u / t_f), flush=True)
    # zero-insertion un-pooling at the five decoder stages of unet_cspn_nyu (B = 24)
    from cspn_monodepth_amd.network import up_pooling as up
    for (H, W, oh, ow, C) in () if args.only else ((8, 10, 15, 19, 1024), (15, 19, 29, 38, 512), (29, 38, 57, 76, 256), (57, 76, 114, 152, 128),
                              (114, 152, 228, 304, 64)):
        x = torch.randn(24, C, H, W, device=dev)
        g = torch.randn(24, C, oh, ow, device=dev)
        wgt = torch.zeros(C, 1, 2, 2, device=dev)
        wgt[:, :, 0, 0] = 1
        with torch.no_grad():
            t_f = timed(lambda: up.up_pooling(x, 2, oh, ow), args.reps)
            t_r = timed(lambda: torch.nn.functional.conv_transpose2d(x, wgt, stride=2, groups=C)[:, :, :oh, :ow], 5)
            t_b = timed(lambda: up._UpPooling.backward(type("c", (), {"geom": (tuple(x.shape), 2, oh, ow)}), g), args.reps)
        fb = (x.numel() + g.numel()) * 4
        row = dict(case="unpool_%dx%d_c%d" % (oh, ow, C), fwd_us=t_f, fwd_GBs=fb / t_f / 1e3, fwd_frac=fb / t_f / 1e3 / PEAK,
                   bwd_us=t_b, bwd_GBs=fb / t_b / 1e3, torch_conv_transpose_us=t_r)
        rows.append(row)
        print("unpool -> %3dx%-3d C=%-4d fwd %7.1f us %6.0f GB/s (%4.1f%%)   bwd %7.1f us %6.0f GB/s   conv_transpose2d formulation %8.1f us (x%.1f)" % (
            oh, ow, C, t_f, row["fwd_GBs"], 100 * row["fwd_frac"], t_b, row["bwd_GBs"], t_r, t_r / t_f), flush=True)
    # Rates above what the HBM can deliver (~6.3 TB/s achievable, 8 TB/s nominal) are Infinity-Cache rates: the working set of such a row (a few
    # tens of MB, re-used by every repetition) sits in the 256 MB cache.  Marked, so that nobody reads them as an HBM fraction.
    for row in rows:
        foot = None
        if "B" in row:
            foot = row["fwd_GBs"] * row["fwd_us"] * 1e3            # = the bytes of one forward
        fast = [k for k in ("fwd_GBs", "bwd_GBs") if row.get(k, 0) > 6300.0]
        row["cache_assisted"] = bool(fast)
        if fast:
            row["cache_assisted_note"] = ("%s above the ~6.3 TB/s the HBM delivers: an Infinity-Cache rate on a working set that fits the 256 MB cache "
                                          "(timed cache-warm), not an HBM fraction" % ", ".join(fast))
        if foot is not None:
            row["fwd_working_set_MB"] = foot / 1e6
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
