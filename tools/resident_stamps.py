#!/usr/bin/env python3
"""Developer tool: per-phase timeline of the resident launch from in-kernel wall-clock stamps (100 MHz)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cspn_monodepth_amd import functional as F
dev = "cuda:0"
B, H, W, T = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (24, 228, 304, 24)))
sparse = len(sys.argv) > 5
g = torch.randn(B, 12, H, W, device=dev)
d = torch.rand(B, H, W, device=dev) * 10
s = d * (torch.rand_like(d) < 0.007) if sparse else None
rp = F.resident_plan(B, H, W, T, int(sparse), 256)
grid = rp["tiles_x"] * rp["tiles_y"] * min(B, rp["images_per_launch"])
st = torch.zeros((grid, 16), dtype=torch.int64, device=dev)
names = ["derive"]
for p in range(-(-T // rp["steps_per_phase"])):
    names += ["stage%d" % p, "steps%d" % p, "xchg%d" % p]
with torch.no_grad():
    for _ in range(5):
        F.forward_resident(g, d, s, T, int(sparse), debug_stamps=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); F.forward_resident(g, d, s, T, int(sparse), debug_stamps=st); e1.record(); e1.synchronize()
t = st.cpu().numpy().astype("float64") / 100.0     # us
t0 = t[:, 0].min()
print("plan", rp, "kernel (events) %.1f us" % (e0.elapsed_time(e1) * 1e3))
print("workgroup start spread: %.2f us" % (t[:, 0].max() - t0))
n = min(len(names) + 1, 16)
for k in range(1, n):
    if t[:, k].max() == 0:
        break
    dt = t[:, k] - t[:, k - 1]
    print("%-8s mean %.2f  min %.2f  max %.2f   (ends at %.2f .. %.2f us after the first start)" % (
        names[k - 1], dt.mean(), dt.min(), dt.max(), t[:, k].min() - t0, t[:, k].max() - t0))
