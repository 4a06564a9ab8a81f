"""Developer probe (round 5): what the guard kernel (cspn_resident_plan.guard) costs the success path of a plain resident inference
call — same box, alternating guard on / off, HIP events around 200 calls each, three repetitions."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
from oracle import c_oracle
c_oracle.build()
DEV = "cuda:0"
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
res = {}
for (B, H, W) in [(24, 228, 304), (3, 228, 304), (1, 352, 1216), (8, 352, 1216)]:
    g, d, s = c_oracle.synthetic_inputs(5, B, H, W, 12, None)
    gt, dt = dev(g), dev(d)
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    rows = []
    for rep in range(3):
        for on in (False, True):
            F.set_resident_guard(on)
            with torch.no_grad():
                for _ in range(30):
                    m(gt, dt)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    m(gt, dt)
                e1.record(); torch.cuda.synchronize()
            rows.append((on, e0.elapsed_time(e1) * 1000 / 200))
    off = sorted(t for on, t in rows if not on); on_ = sorted(t for on, t in rows if on)
    res["%dx%dx%d" % (B, H, W)] = dict(us_per_call_guard_off=off, us_per_call_guard_on=on_, delta_median_us=on_[1] - off[1])
    print(B, H, W, res["%dx%dx%d" % (B, H, W)], flush=True)
F.ensure_resident_ok()
os.makedirs(os.path.join(ROOT, "gpurun_out", "r05_guard"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r05_guard", "guard_price.json"), "w"), indent=1)
