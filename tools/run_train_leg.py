#!/usr/bin/env python3
"""Developer tool: N forward+backward passes of the CSPN module and nothing else — the command rocprofv3 wraps to profile
the training-shaped kernels.  Default: the 3x3 module at config 2 (history-keeping forward, reverse sweep, cspn_grad_tail);
--K 5 [--dtype f16]: the K x K softmax module at config 3's shape (softmax prepare, history-keeping launches, transposed
launches, the PAC backward tail)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
import cspn_monodepth_amd as pkg               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=24)
ap.add_argument("--H", type=int, default=228)
ap.add_argument("--W", type=int, default=304)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--sparse", action="store_true")
ap.add_argument("--T", type=int, default=0, help="propagation steps (default: 24 for K=3, 12 for K=5)")
ap.add_argument("--K", type=int, default=3)
ap.add_argument("--dtype", choices=("f32", "f16"), default="f32")
ap.add_argument("--module", choices=("auto", "new", "ours"), default="auto", help="K = 3: CSPN_new (default) or CSPN_ours (softmax taps)")
ap.add_argument("--state", choices=("reference", "input"), default="reference",
                help="CSPN_ours with half inputs: fp32 state as the reference promotes it (default) or the input dtype (what bench.py --workload pac5 runs)")
a = ap.parse_args()
dev = "cuda:0"
dt = torch.float16 if a.dtype == "f16" else torch.float32
T = a.T or (24 if a.K == 3 else 12)
ours = a.module == "ours" or (a.module == "auto" and a.K != 3)
C = a.K * a.K - 1 if ours else 12
g = torch.randn(a.batch, C, a.H, a.W, device=dev).to(dt).requires_grad_(True)
d = (torch.rand(a.batch, 1, a.H, a.W, device=dev) * 10).to(dt).requires_grad_(True)
s = (d.detach() * (torch.rand_like(d) < 0.007)) if a.sparse else None
cot = torch.randn(a.batch, 1, a.H, a.W, device=dev).to(dt)
if not ours:
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    run = lambda: m(g, d, s)                   # noqa: E731
else:
    m = pkg.CSPN_ours.AffinityPropagate(T, state_dtype="reference" if a.state == "reference" else None)
    run = lambda: m(d, g, s)                   # noqa: E731
for _ in range(a.iters):
    g.grad = None
    d.grad = None
    run().backward(cot)
torch.cuda.synchronize()
print("done", a.iters)
