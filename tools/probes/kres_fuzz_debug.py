#!/usr/bin/env python3
"""Developer probe: replay the seeded fuzz of tests/test_hip_kres.py and print every mismatching case with its geometry."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np                                   # noqa: E402
import torch                                         # noqa: E402
from cspn_monodepth_amd import functional as F       # noqa: E402
from oracle import c_oracle                          # noqa: E402
import test_hip_kres as tk                           # noqa: E402

bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(4200 + seed)
    for _ in range(40):
        K = int(rng.choice([3, 5])); f32 = bool(rng.random() < 0.4); B = int(rng.integers(1, 7)); H = int(rng.integers(1, 90))
        W = int(8 * rng.integers(1, 20)); T = int(rng.integers(1, 14)); S = int(rng.choice([0, 2, 4, 6, 8]))
        threads = int(rng.choice([0, 512, 768])); sparse = bool(rng.random() < 0.5)
        gdt = F.CSPN_F32 if f32 else F.CSPN_F16
        rp = F.kres_plan(K, B, H, W, T, int(sparse), 0, S, threads, gdt)
        if rp is None:
            continue
        x, gd, s = tk.inputs(c_oracle, B, H, W, K, sparse, seed=70 + seed)
        tdt = torch.float32 if f32 else torch.float16
        xt, gt, st = tk.dev(x, tdt), tk.dev(gd, tdt), tk.dev(s, tdt)
        state = None if (f32 or rng.random() < 0.5) else torch.float32
        sdt = tdt if state is None else state
        ref = tk.multi_launch(xt, gt, st, T, rp["steps_per_phase"], state)[:, 0]
        with torch.no_grad():
            out = F.pac_forward_resident(gt, xt[:, 0].to(sdt).contiguous(), None if st is None else st[:, 0].to(sdt).contiguous(), T,
                                         steps_per_phase=S, threads=threads)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            d = (out.float() - ref.float()).abs()
            idx = (out != ref).nonzero()
            print("MISMATCH K=%d f32=%s B=%d %dx%d T=%d S=%d(->%d) threads=%d sparse=%s state=%s plan=%s: n=%d max=%.4g nan_out=%d rows %d..%d cols %d..%d first %s" % (
                K, f32, B, H, W, T, S, rp["steps_per_phase"], threads, sparse, state,
                {k: rp[k] for k in ("tiles_x", "tiles_y", "tile_w", "tile_h", "quads_per_thread", "threads", "images_per_launch", "launches")},
                idx.shape[0], float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else -1, int(torch.isnan(out).sum()),
                int(idx[:, 1].min()), int(idx[:, 1].max()), int(idx[:, 2].min()), int(idx[:, 2].max()), idx[:4].tolist()))
print("mismatching cases:", bad)
try:
    F.ensure_resident_ok()
except RuntimeError as e:
    print("resident error:", e)
