"""Developer probe (round 5): per-phase timeline of the K = 5 reverse sweep (cspnk_resident<TRANS>, one launch of two rounds at config 3's
shape) from in-kernel wall-clock stamps (100 MHz) of round 0's workgroups.  usage: r05_sweep_stamps.py [B H W T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cspn_monodepth_amd import functional as F
dev = "cuda:0"
B, H, W, T = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (24, 228, 304, 12)))
torch.manual_seed(0)
g = torch.randn(B, 24, H, W, device=dev).half()
cot = torch.randn(B, H, W, device=dev).half()
with torch.no_grad():
    wk, _ = F.pac_prepare(g)
rp = F.kres_plan(5, B, H, W, T, 0)
grid = rp["tiles_x"] * rp["tiles_y"] * min(B, rp["images_per_launch"])
st = torch.zeros((grid, 16), dtype=torch.int64, device=dev)
names = ["parked", "gather"]
for p in range(-(-T // rp["steps_per_phase"])):
    names += ["stage%d" % p, "steps%d" % p, "xchg%d" % p]
names[-1] = "epilogue"
with torch.no_grad():
    for _ in range(3):
        F.pac_transposed_resident(wk, cot, None, T, debug_stamps=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); F.pac_transposed_resident(wk, cot, None, T, debug_stamps=st); e1.record(); e1.synchronize()
t = st.cpu().numpy().astype("float64") / 100.0
t0 = t[:, 0].min()
print("plan", {k: rp[k] for k in ("steps_per_phase", "tiles_x", "tiles_y", "tile_w", "tile_h", "threads", "images_per_launch", "launches")},
      "call (events, with stamps and guard) %.1f us; stamps: round 0 of %d rounds" % (e0.elapsed_time(e1) * 1e3, -(-B // rp["images_per_launch"])))
print("workgroup start spread %.2f us" % (t[:, 0].max() - t0))
for k in range(1, 16):
    if t[:, k].max() == 0 or k > len(names):
        break
    dt = t[:, k] - t[:, k - 1]
    print("  %-9s mean %.2f  min %.2f  max %.2f   (ends %.2f .. %.2f us)" % (names[k - 1], dt.mean(), dt.min(), dt.max(), t[:, k].min() - t0, t[:, k].max() - t0))
