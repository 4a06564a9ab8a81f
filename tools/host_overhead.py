#!/usr/bin/env python3
"""Host-side cost of the Python wrappers (no GPU sync inside the timed calls) — developer micro-benchmark."""
import os, sys, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
B, H, W, T = 3, 228, 304, 24
g = torch.randn(B, 12, H, W, device="cuda", requires_grad=True)
d = (torch.rand(B, 1, H, W, device="cuda") * 10).requires_grad_(True)
cot = torch.randn(B, 1, H, W, device="cuda")
m = pkg.CSPN_new.AffinityPropagate(T, 3)
def fb():
    g.grad = None; d.grad = None
    m(g, d).backward(cot)
for _ in range(10): fb()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(200): fb()
t1 = time.perf_counter()
pr.disable(); torch.cuda.synchronize()
print("host time per fwd+bwd: %.1f us" % ((t1 - t0) / 200 * 1e6))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3500])
