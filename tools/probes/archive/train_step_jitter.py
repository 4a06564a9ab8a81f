#!/usr/bin/env python3
"""Developer probe: per-pass wall time of the training-shaped step (forward with history + backward) at config 2 — where do the
occasional slow passes of bench.py's `training_step` leg come from?  Each pass is timed with a device synchronisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cspn_monodepth_amd as pkg
dev = "cuda"
B, H, W, T = 24, 228, 304, 24
g = torch.randn(B, 12, H, W, device=dev, requires_grad=True)
d = (torch.rand(B, 1, H, W, device=dev) * 10).requires_grad_(True)
cot = torch.randn(B, 1, H, W, device=dev)
m = pkg.CSPN_new.AffinityPropagate(T, 3)
def fb():
    g.grad = None; d.grad = None
    m(g, d, None).backward(cot)
for _ in range(10): fb()
torch.cuda.synchronize()
for mode in ("sync every pass", "free running, 30 passes x 6"):
    ts = []
    if mode.startswith("sync"):
        for _ in range(60):
            t0 = time.perf_counter(); fb(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    else:
        for _ in range(6):
            t0 = time.perf_counter()
            for _ in range(30): fb()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 30 * 1e6)
    print(mode, " ".join("%.0f" % t for t in ts))
print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.peak", "allocated_bytes.all.peak")})
