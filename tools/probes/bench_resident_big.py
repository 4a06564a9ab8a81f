#!/usr/bin/env python3
"""Developer tool: resident vs multi-launch for batches that need 4-8 resident launches (scored forward, us)."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
dev = "cuda:0"
for (B, H, W) in ((96, 228, 304), (192, 228, 304), (32, 352, 1216)):
    g = torch.randn(B, 12, H, W, device=dev); d = torch.rand(B, 1, H, W, device=dev) * 10
    tg = d + 0.1; m = pkg.CSPN_new.AffinityPropagate(24, 3); acc = pkg.evaluation.new_accumulator(dev)
    res = {}
    for mode in ("off", "on"):
        F.set_resident(mode)
        with torch.no_grad():
            for _ in range(5): m.forward_scored(g, d, None, tg, acc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): m.forward_scored(g, d, None, tg, acc)
            e1.record(); e1.synchronize()
            res[mode] = e0.elapsed_time(e1) * 1e3 / 20
    print(B, H, W, res, F.resident_plan(B, H, W, 24, 0, 256)["launches"], flush=True)
