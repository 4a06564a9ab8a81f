"""Developer probe: host time of ENQUEUEING the resident launches by form (inference / training forward / reverse sweep) — the
training forms took 30-70 us per launch call against ~5 us for a streaming kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cspn_monodepth_amd import functional as F
dev = "cuda:0"
B, H, W, T = 24, 228, 304, 24
g = torch.randn(B, 12, H, W, device=dev); d = torch.rand(B, H, W, device=dev) * 10
w8 = torch.rand(B, 8, H, W, device=dev); gT = torch.randn(B, H, W, device=dev)
def bench(name, fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tw = time.perf_counter() - t0
    print("%-28s host %.1f us per call, wall %.1f us per call" % (name, th / n * 1e6, tw / n * 1e6))
with torch.no_grad():
    bench("inference", lambda: F.forward_resident(g, d, None, T, 0))
    bench("training forward (history)", lambda: F.forward_resident(g, d, None, T, 0, keep_history=True))
    bench("reverse sweep", lambda: F.transposed_resident(w8, gT, None, T))
    def both():
        F.forward_resident(g, d, None, T, 0, keep_history=True); F.transposed_resident(w8, gT, None, T)
    bench("forward + reverse alternating", both)

# ---- where inside one inference call does the host time go?
import collections
L0 = F._lib.lib()
acc = collections.OrderedDict()
c0 = L0.cspn3_forward_resident
def ccall(*a):
    t = time.perf_counter(); r = c0(*a); acc["C call"] = acc.get("C call", 0.0) + time.perf_counter() - t; return r
class LW(object):
    def __getattr__(self, k): return ccall if k == "cspn3_forward_resident" else getattr(L0, k)
lw = LW(); F._lib.lib = lambda: lw
e0 = torch.empty
def emp(*a, **k):
    t = time.perf_counter(); r = e0(*a, **k); acc["torch.empty"] = acc.get("torch.empty", 0.0) + time.perf_counter() - t; return r
torch.empty = emp
with torch.no_grad():
    for (BB, HH, WW) in ((24, 228, 304), (1, 64, 64)):
        gg = torch.randn(BB, 12, HH, WW, device=dev); dd = torch.rand(BB, HH, WW, device=dev) * 10
        for sync_each in (False, True):
            for _ in range(5): F.forward_resident(gg, dd, None, T, 0)
            torch.cuda.synchronize(); acc.clear()
            t0 = time.perf_counter()
            for _ in range(200):
                F.forward_resident(gg, dd, None, T, 0)
                if sync_each: torch.cuda.synchronize()
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            print("B=%d %dx%d sync_each=%s: host %.1f us per call; %s" % (BB, HH, WW, sync_each, th / 200 * 1e6, {k: round(v / 200 * 1e6, 1) for k, v in acc.items()}))
