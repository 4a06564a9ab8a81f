#!/usr/bin/env python3
"""round 6: host issue time of config 3's scored forward (CSPN_ours.forward_scored -> pac_refine_and_score -> cspnk_forward_resident) against its
GPU time: is the bench line of config 3 host-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import evaluation as ev
torch.manual_seed(0)
for B in (24, 3):
    H, W = 228, 304
    gd = torch.randn(B, 24, H, W, device="cuda").half(); x = (torch.rand(B, 1, H, W, device="cuda") * 10).half(); tg = (x.float() + 0.1).half()
    m = pkg.CSPN_ours.AffinityPropagate(12, state_dtype=None); acc = ev.new_accumulator("cuda")
    with torch.no_grad():
        for _ in range(30): m.forward_scored(x, gd, None, tg, acc)
        torch.cuda.synchronize()
        for n in (20, 200):
            t0 = time.perf_counter()
            for _ in range(n): m.forward_scored(x, gd, None, tg, acc)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("B=%d n=%d: host issue %.1f us per call, wall %.1f us per call" % (B, n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
