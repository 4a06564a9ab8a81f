#!/usr/bin/env python3
"""Developer probe: seeded sweep of the TRAINING forms and the SCORED forms of the resident kernels against the multi-launch
schedule (bit for bit): history planes, published weights / S / softmax taps, reverse sweep histories, fused metric sums.
    python tools/probes/resident_train_fuzz.py [seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                                   # noqa: E402
import torch                                         # noqa: E402
import cspn_monodepth_amd as pkg                     # noqa: E402
from cspn_monodepth_amd import functional as F       # noqa: E402
from cspn_monodepth_amd import evaluation as ev      # noqa: E402

dev = "cuda:0"
bad = n = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    rng = np.random.default_rng(7700 + seed)
    torch.manual_seed(seed)
    for _ in range(12):
        B, H, W, T = int(rng.integers(1, 7)), int(rng.integers(1, 100)), int(8 * rng.integers(1, 24)), int(rng.integers(1, 26))
        sparse = bool(rng.random() < 0.5)
        if F.resident_plan(B, H, W, T, int(sparse), 0) is None:
            continue
        d = torch.rand(B, H, W, device=dev) * 10
        s = d * (torch.rand(B, H, W, device=dev) < 0.05) if sparse else None
        tg = (d + 0.1 * torch.randn(B, H, W, device=dev)).clamp_min(0)
        blend = F.BLEND_SPARSE if sparse else F.BLEND_NONE
        case = (B, H, W, T, sparse)
        with torch.no_grad():
            # --- CSPN_new: training forward, reverse sweep, scored forward
            g = torch.randn(B, 12, H, W, device=dev)
            _, h0, w0, S0 = F.propagate_from_guidance(g, d, s, T, blend, keep_history=True, return_weights=True)
            _, h1, w1, S1 = F.forward_resident(g, d, s, T, int(sparse), keep_history=True)
            ok = torch.equal(h0, h1) and torch.equal(w0, w1) and torch.equal(S0, S1)
            gT = torch.randn(B, H, W, device=dev)
            prev = F._RESIDENT_MODE
            F.set_resident("off"); _, gh0 = F._reverse_sweep(w0, 3, T, s, gT, None)
            F.set_resident("on"); _, gh1 = F._reverse_sweep(w0, 3, T, s, gT, None)
            ok = ok and torch.equal(gh0, gh1)
            # the volume-free forms (ABI 9): S-only training forward, reverse sweep rebuilt from guidance + S
            _, h2, w2, S2 = F.forward_resident(g, d, s, T, int(sparse), keep_history=True, publish_weights=False)
            gh2 = F.transposed_resident_guidance(g, S2, gT, s, T)
            ok = ok and w2 is None and torch.equal(h0, h2) and torch.equal(S0, S2) and torch.equal(gh0, gh2)
            a0, a1 = ev.new_accumulator(dev), ev.new_accumulator(dev)
            m = pkg.CSPN_new.AffinityPropagate(T, 3)
            o1 = m.forward_scored(g, d.unsqueeze(1), None if s is None else s.unsqueeze(1), tg.unsqueeze(1), a1)
            F.set_resident("off")
            o0 = m.forward_scored(g, d.unsqueeze(1), None if s is None else s.unsqueeze(1), tg.unsqueeze(1), a0)
            F.set_resident(prev)
            ok = ok and torch.equal(o0, o1) and torch.allclose(a0.sum(0), a1.sum(0), rtol=1e-6)
            # --- CSPN_ours K = 3 fp32: training forward, scored forward
            gp = torch.randn(B, 8, H, W, device=dev)
            wk0, _ = F.pac_prepare(gp)
            _, ph0 = F.propagate(wk0, d, s, 3, T, blend, keep_history=True)
            _, ph1, wk1 = F.pac_forward_resident_history(gp, d, s, T)
            ok = ok and torch.equal(ph0, ph1) and torch.equal(wk0, wk1)
            mo = pkg.CSPN_ours.AffinityPropagate(T)
            a2, a3 = ev.new_accumulator(dev), ev.new_accumulator(dev)
            F.set_resident("on")
            p1 = mo.forward_scored(d.unsqueeze(1), gp, None if s is None else s.unsqueeze(1), tg.unsqueeze(1), a3)
            F.set_resident("off")
            p0 = mo.forward_scored(d.unsqueeze(1), gp, None if s is None else s.unsqueeze(1), tg.unsqueeze(1), a2)
            F.set_resident(prev)
            # (the multi-launch default plan may use another phase length: fp32 results are plan-independent)
            ok = ok and torch.equal(p0, p1) and torch.allclose(a2.sum(0), a3.sum(0), rtol=1e-6)
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", case, flush=True)
F.ensure_resident_ok()
print("cases %d, mismatching %d" % (n, bad))
sys.exit(1 if bad else 0)
