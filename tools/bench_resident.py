#!/usr/bin/env python3
"""Developer tool: resident single-launch forward vs the multi-launch schedule (HIP events, same tensors)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F

dev = "cuda:0"
rows = []
for (B, H, W) in ((24, 228, 304), (12, 228, 304), (6, 228, 304), (3, 228, 304), (1, 228, 304), (8, 352, 1216), (4, 352, 1216),
                  (2, 352, 1216), (1, 352, 1216), (48, 228, 304)):
    g = torch.randn(B, 12, H, W, device=dev)
    d = torch.rand(B, 1, H, W, device=dev) * 10
    s = d * (torch.rand_like(d) < 0.007)
    tg = d + 0.1
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    acc = pkg.evaluation.new_accumulator(dev)
    for sparse in (None, s):
        res = {}
        for mode, S in (("off", 0), ("on", 8), ("on", 6), ("on", 4)):
            F.set_resident(mode)
            if mode == "on" and F.resident_plan(B, H, W, 24, int(sparse is not None), 256, S) is None:
                continue

            def run():
                if mode == "on":
                    return F.forward_resident(g, d[:, 0], None if sparse is None else sparse[:, 0], 24, int(sparse is not None),
                                              score=(tg[:, 0], acc), steps_per_phase=S)
                return m.forward_scored(g, d, sparse, tg, acc)
            with torch.no_grad():
                for _ in range(10):
                    run()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(50):
                        run()
                    e1.record(); e1.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
            res["%s%d" % (mode, S)] = round(best, 2)
        F.check_resident_errors()
        rp = F.resident_plan(B, H, W, 24, int(sparse is not None), 256)
        rows.append(dict(B=B, H=H, W=W, sparse=sparse is not None, us=res, plan=rp))
        print(json.dumps(rows[-1]), flush=True)
F.set_resident("auto")
