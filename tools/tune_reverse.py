#!/usr/bin/env python3
"""Plan sweep for the reverse sweep (transposed recurrence with history) — developer tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cspn_monodepth_amd import functional as F
from tools.tune import timed
from bench import WORKLOADS, make_inputs
wl = dict(WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "nyu"])
B, H, W, T = wl["B"], wl["H"], wl["W"], wl["T"]
g, d, s, _ = make_inputs(wl, B, torch.device("cuda", 0), 1, False)
w8, _, _ = F.cspn3_prepare(g)
cot = torch.randn(B, H, W, device="cuda")
print("default: %.1f us" % timed(lambda: F._reverse_sweep(w8, 3, T, None, cot, None), 10))
d0 = d[:, 0].contiguous()
print("forward with history, default: %.1f us" % timed(lambda: F.propagate(w8, d0, None, 3, T, F.BLEND_NONE, keep_history=True), 10))
rows = []
for plan in F.candidate_plans(3, H, W, T):
    try:
        p = F.resolve_plan(3, B, H, W, T, True, plan)
        if (p["quads_per_thread"], p["threads"]) not in F._TRANSPOSED_INSTANCES[3]:
            continue
        rows.append((timed(lambda: F._reverse_sweep(w8, 3, T, None, cot, plan), 6, 1),
                     timed(lambda: F.propagate(w8, d0, None, 3, T, F.BLEND_NONE, keep_history=True, plan=plan), 6, 1), plan))
    except RuntimeError:
        continue
for r in sorted(rows, key=lambda r: r[0])[:6]:
    print("reverse %.1f us (forward+history %.1f us)  %s" % r)
for r in sorted(rows, key=lambda r: r[1])[:3]:
    print("forward+history %.1f us (reverse %.1f us)  %s" % (r[1], r[0], r[2]))
