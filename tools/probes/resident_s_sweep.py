#!/usr/bin/env python3
"""Developer probe: the K = 3 resident launch over phase lengths S (incl. 12 and 24 = no exchange at all) at per-GPU shard
sizes: is the fixed cost of small shards the exchanges?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                         # noqa: E402
from cspn_monodepth_amd import functional as F       # noqa: E402

dev = "cuda:0"
T = 24
for (B, H, W) in ((3, 228, 304), (1, 352, 1216), (1, 228, 304), (6, 228, 304), (2, 352, 1216), (24, 228, 304)):
    g = torch.randn(B, 12, H, W, device=dev)
    d = torch.rand(B, H, W, device=dev) * 10
    for S in (4, 6, 8, 12, 24):
        rp = F.resident_plan(B, H, W, T, 0, 256, S)
        if rp is None:
            print("B=%d %dx%d S=%d: no plan" % (B, H, W, S))
            continue
        with torch.no_grad():
            for _ in range(5):
                F.forward_resident(g, d, None, T, 0, steps_per_phase=S)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    F.forward_resident(g, d, None, T, 0, steps_per_phase=S)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 30)
        print("B=%d %dx%d S=%2d: %6.2f us  (nq %d, tiles %dx%d of %dx%d, launches %d, region/tile %.2f)" % (
            B, H, W, S, best, rp["quads_per_thread"], rp["tiles_x"], rp["tiles_y"], rp["tile_w"], rp["tile_h"], rp["launches"],
            rp["region_over_tile"]), flush=True)
F.ensure_resident_ok()
