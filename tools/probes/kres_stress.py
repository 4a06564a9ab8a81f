#!/usr/bin/env python3
"""Developer tool: hammer the K x K resident launches (fp16 and fp32 guidance, both workgroup sizes, ±sparse) and compare
every output with the multi-launch result bit for bit; inputs alternate so that stale exchange data of the previous call
would show up."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                         # noqa: E402
import cspn_monodepth_amd as pkg                     # noqa: E402
from cspn_monodepth_amd import functional as F       # noqa: E402

dev = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
torch.manual_seed(0)
bad_total = 0
# form "fma": the bits of the multi-launch schedule are the reference.  form "dot2" (what "auto" picks for K = 5 with fp16 guidance
# and planes since round 4): not those bits by design (the state is rounded to half after every step) — the reference is the FIRST
# resident output of each input set (a stale exchange plane or a race would make a later launch differ from it) and the
# multi-launch result must lie within the fp16 tolerance of the configuration (4e-3 of the depth range).
form = sys.argv[2] if len(sys.argv) > 2 else "fma"
F.set_kres_step_form(form)
for (K, T, B, H, W, dt, sparse) in ((5, 12, 24, 228, 304, torch.float16, False), (5, 12, 3, 228, 304, torch.float16, True),
                                    (3, 24, 24, 228, 304, torch.float32, True), (5, 12, 1, 352, 1216, torch.float16, False),
                                    (3, 24, 6, 120, 160, torch.float16, False)):
    sets = []
    for k in range(3):
        g = torch.randn(B, K * K - 1, H, W, device=dev).to(dt)
        d = (torch.rand(B, 1, H, W, device=dev) * 10).to(dt)
        s = (d * (torch.rand(B, 1, H, W, device=dev) < 0.01)).to(dt) if sparse else None
        sets.append((d, g, s))
    m = pkg.CSPN_ours.AffinityPropagate(T, state_dtype=None)
    with torch.no_grad():
        F.set_resident("on")
        rp = F.pac_resident_supported(sets[0][1], sets[0][0][:, 0].contiguous(), None if not sparse else sets[0][2][:, 0].contiguous(), T)
        F.set_resident("off")
        mref = pkg.CSPN_ours.AffinityPropagate(T, plan=dict(steps_per_launch=rp["steps_per_phase"]), state_dtype=None)
        refs = [mref(d, g, sparse_depth=s) for d, g, s in sets]
        F.set_resident("on")
        if form != "fma":
            firsts = [m(d, g, sparse_depth=s) for d, g, s in sets]
            for f, r in zip(firsts, refs):
                assert float((f.float() - r.float()).abs().max()) <= 4e-3 * 10 * 2, "dot-product form outside the fp16 tolerance"
            refs = firsts
        bad, outs = 0, []
        for it in range(iters):
            k = it % 3
            d, g, s = sets[k]
            outs.append((k, m(d, g, sparse_depth=s)))
            if len(outs) == 15:
                for k2, o in outs:
                    if not torch.equal(o, refs[k2]):
                        bad += 1
                        if bad <= 3:
                            idx = (o != refs[k2]).nonzero()
                            print("  mismatch at iter ~%d: %d px, first %s, last %s" % (it, idx.shape[0], idx[0].tolist(), idx[-1].tolist()))
                outs = []
    F.ensure_resident_ok()
    print("[%s] K=%d T=%d B=%d %dx%d %s sparse=%s plan S=%d threads=%d launches=%d: %d / %d mismatching outputs" % (
        form, K, T, B, H, W, str(dt).split(".")[-1], sparse, rp["steps_per_phase"], rp["threads"], rp["launches"], bad, iters), flush=True)
    bad_total += bad
F.set_resident("auto")
sys.exit(1 if bad_total else 0)
