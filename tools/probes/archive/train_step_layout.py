#!/usr/bin/env python3
"""Developer probe: does the training-shaped step's time depend on WHERE the caching allocator puts its planes?  A pad tensor of a
varying size is allocated first (it shifts every later block), then 60 free-running passes are timed; per-kernel times come from
HIP events around the three launches of one pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cspn_monodepth_amd as pkg
dev = "cuda"
B, H, W, T = 24, 228, 304, 24
m = pkg.CSPN_new.AffinityPropagate(T, 3)
for pad_mb in (0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 64, 100, 128, 200):
    torch.cuda.empty_cache()
    pad = torch.empty(pad_mb * (1 << 20), dtype=torch.uint8, device=dev) if pad_mb else None
    g = torch.randn(B, 12, H, W, device=dev, requires_grad=True)
    d = (torch.rand(B, 1, H, W, device=dev) * 10).requires_grad_(True)
    cot = torch.randn(B, 1, H, W, device=dev)
    def fb():
        g.grad = None; d.grad = None
        m(g, d, None).backward(cot)
    for _ in range(10): fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60): fb()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 60 * 1e6
    print("pad %4d MB: %.1f us per pass   (g at +%d KB mod 2 MB)" % (pad_mb, dt, (g.data_ptr() % (2 << 20)) >> 10), flush=True)
    del g, d, cot, pad
