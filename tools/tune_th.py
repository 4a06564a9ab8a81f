#!/usr/bin/env python3
"""Finer sweep of tile heights for a fixed (S, nq, threads): does a balanced tile count beat the largest tile?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cspn_monodepth_amd import functional as F
from tools.tune import timed
from bench import WORKLOADS, make_inputs
wl = dict(WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "nyu"])
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, H, W, T = wl["B"], wl["H"], wl["W"], wl["T"]
g, d, s, _ = make_inputs(wl, B, torch.device("cuda", 0), 1, False)
w, _, _ = F.cspn3_prepare(g)
d0 = d[:, 0].contiguous()
rows = []
for threads, nq in ((1024, 1), (512, 1), (1024, 2)):
    for tw in range(24, 161, 4):
        hx = -(-(S - 1) // 4) * 4
        wq = (tw + 2 * hx) // 4
        if wq > threads: continue
        th_max = nq * (threads // wq) - 2 * (S - 1)
        for th in range(8, min(th_max, H) + 1):
            if -(-H // th) == -(-H // (th + 1)) and th != th_max: continue      # only heights that change the tile count
            plan = dict(steps_per_launch=S, tile_w=tw, tile_h=th, quads_per_thread=nq, threads=threads)
            try:
                F.resolve_plan(3, B, H, W, T, False, plan)
                us = timed(lambda: F.propagate(w, d0, None, 3, T, F.BLEND_NONE, plan=plan), 6, 1)
            except RuntimeError:
                continue
            tiles = B * -(-W // tw) * -(-H // th)
            rows.append((us, tw, th, nq, threads, tiles))
rows.sort()
for r in rows[:15]:
    print("%.1f us  tile %dx%d nq=%d thr=%d tiles=%d (%.2f rounds of %d)" % (r[0], r[1], r[2], r[3], r[4], r[5], r[5] / (256 * (2048 // r[4]) / (2 if r[3] == 2 else 1)), 256 * (2048 // r[4])))
