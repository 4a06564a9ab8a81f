#!/usr/bin/env python3
"""Raw-KITTI width (1242, W % 4 == 2): row-padded quad kernels vs generic per-pixel kernels (developer micro-benchmark)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_monodepth_amd as pkg
from tools.tune import timed
B, H, W = 8, 375, 1242
g = torch.randn(B, 12, H, W, device="cuda"); d = torch.rand(B, 1, H, W, device="cuda") * 10
with torch.no_grad():
    print("row-padded quad kernels : %.1f us" % timed(lambda: pkg.CSPN_new.AffinityPropagate(24, 3)(g, d), 10))
    print("generic per-pixel kernels: %.1f us" % timed(lambda: pkg.CSPN_new.AffinityPropagate(24, 3, plan=dict(force_scalar=1))(g, d), 5))
    g4, d4 = g[..., :1240].contiguous(), d[..., :1240].contiguous()
    print("W=1240 (no padding needed): %.1f us" % timed(lambda: pkg.CSPN_new.AffinityPropagate(24, 3)(g4, d4), 10))
