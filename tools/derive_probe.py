import sys, torch
sys.path.insert(0, '/root/repo')
from cspn_monodepth_amd import functional as F
from tools.tune import timed
from bench import WORKLOADS, make_inputs
wl = dict(WORKLOADS["nyu"]); B,H,W,T = wl["B"],wl["H"],wl["W"],wl["T"]
g, d, s, _ = make_inputs(wl, B, torch.device("cuda",0), 1, False)
d0 = d[:,0].contiguous()
w,_,_ = F.cspn3_prepare(g)
for TT in (8, 16, 24):
    a = timed(lambda: F.propagate(w, d0, None, 3, TT, F.BLEND_NONE), 20)
    b = timed(lambda: F.propagate_from_guidance(g, d0, None, TT, F.BLEND_NONE), 20)
    c = timed(lambda: F.propagate_from_guidance(g, d0, None, TT, F.BLEND_NONE, publish_weights=False), 20)
    print("T=%d: prepared-weights %.1f us | derive(+publish if >1 launch) %.1f us | derive every launch %.1f us" % (TT, a, b, c))
print("prepare alone %.1f us" % timed(lambda: F.cspn3_prepare(g), 20))
