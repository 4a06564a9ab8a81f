#!/usr/bin/env python3
"""Developer tool: N forward+backward passes of the 3x3 module at config 2 (or --batch/--hw), nothing else — the command
rocprofv3 wraps to profile the training-shaped kernels (history-keeping forward, WSRC=2 reverse sweep, cspn_grad_tail)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cspn_monodepth_amd as pkg
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=24); ap.add_argument("--H", type=int, default=228); ap.add_argument("--W", type=int, default=304)
ap.add_argument("--iters", type=int, default=12); ap.add_argument("--sparse", action="store_true"); ap.add_argument("--T", type=int, default=24)
a = ap.parse_args()
dev = "cuda:0"
g = torch.randn(a.batch, 12, a.H, a.W, device=dev, requires_grad=True)
d = (torch.rand(a.batch, 1, a.H, a.W, device=dev) * 10).requires_grad_(True)
s = (d.detach() * (torch.rand_like(d) < 0.007)) if a.sparse else None
cot = torch.randn(a.batch, 1, a.H, a.W, device=dev)
m = pkg.CSPN_new.AffinityPropagate(a.T, 3)
for _ in range(a.iters):
    g.grad = None; d.grad = None
    m(g, d, s).backward(cot)
torch.cuda.synchronize()
print("done", a.iters)
