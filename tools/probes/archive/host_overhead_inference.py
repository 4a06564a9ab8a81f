#!/usr/bin/env python3
"""Host-side cost per inference step (module.forward_scored) with a tiny batch so the GPU never back-pressures."""
import os, sys, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
B, H, W, T = 1, 64, 64, 24
g = torch.randn(B, 12, H, W, device="cuda")
d = torch.rand(B, 1, H, W, device="cuda") * 10
t = torch.rand(B, 1, H, W, device="cuda") * 10
m = pkg.CSPN_new.AffinityPropagate(T, 3)
acc = pkg.evaluation.new_accumulator(g.device)
ev = F.EventLog(4000)
with torch.no_grad():
    for log in (None, ev):
        F.set_event_log(log)
        for _ in range(50): m.forward_scored(g, d, None, t, acc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000): m.forward_scored(g, d, None, t, acc)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print("host time per forward_scored (event log %s): %.1f us" % ("on" if log else "off", (t1 - t0) / 1000 * 1e6))
    F.set_event_log(None)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(1000): m.forward_scored(g, d, None, t, acc)
    pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
