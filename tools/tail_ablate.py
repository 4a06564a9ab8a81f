#!/usr/bin/env python3
"""Timing of the backward-tail pieces (developer micro-benchmark)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cspn_monodepth_amd import functional as F
from tools.tune import timed
B, H, W, T = 24, 228, 304, 24
dev = "cuda"
g = torch.randn(B, 12, H, W, device=dev)
d0 = torch.rand(B, H, W, device=dev) * 10
cot = torch.randn(B, 1, H, W, device=dev)
w8, S, _ = F.cspn3_prepare(g, want_s=True)
_, hist = F.propagate(w8, d0, None, 3, T, F.BLEND_NONE, keep_history=True)
print("reverse sweep (transpose + 24 steps with history): %.1f us" % timed(lambda: F._reverse_sweep(w8, 3, T, None, cot, None), 20))
g_T, ghist = F._reverse_sweep(w8, 3, T, None, cot, None)
print("grad_weights (tail variant 0): %.1f us" % timed(lambda: F._grad_weights(w8, 3, T, d0, hist, None, g_T, ghist), 20))
L, P, st = F._lib.lib(), F._p, F._stream(g.device)
gg = torch.empty_like(g); gd0 = torch.empty_like(d0)
print("cspn3_backward_tail (variant 1): %.1f us" % timed(lambda: L.cspn3_backward_tail(P(d0), P(hist), P(g_T), P(ghist), None, P(g), g.stride(0), g.stride(1), 12, P(w8), P(S), P(gg), P(gd0), 0, B, H, W, T, st), 20))
for TT in (1, 4, 12):
    print("  tail with T=%d: %.1f us" % (TT, timed(lambda: L.cspn3_backward_tail(P(d0), P(hist), P(g_T), P(ghist), None, P(g), g.stride(0), g.stride(1), 12, P(w8), P(S), P(gg), P(gd0), 0, B, H, W, TT, st), 20)))
print("transpose: %.1f us" % timed(lambda: F.transpose_weights(w8, 3, H, W), 20))
