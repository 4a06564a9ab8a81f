#!/bin/bash
# seeded sweeps over seeds the suite does not use: the volume-free training forms (ABI 9) and the stride-2 PAC kernels; then a
# stress loop of training steps (volume-free reverse sweep, alternating inputs) against the multi-launch backward
O=gpurun_out/fuzz_new; mkdir -p $O
timeout 1500 python tools/probes/resident_train_fuzz.py 40 > $O/train_fuzz.txt 2>&1; tail -3 $O/train_fuzz.txt
timeout 1500 python tools/probes/pac_fuzz_more.py 3000 3300 > $O/pac_fuzz.txt 2>&1; tail -3 $O/pac_fuzz.txt
timeout 900 python - > $O/train_stress.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
dev = "cuda:0"
torch.manual_seed(1)
bad = n = 0
for (B, H, W, sparse) in ((24, 228, 304, False), (24, 228, 304, True), (3, 228, 304, True), (1, 352, 1216, False)):
    sets = []
    for k in range(3):
        g = torch.randn(B, 12, H, W, device=dev); d = torch.rand(B, 1, H, W, device=dev) * 10
        sp = d * (torch.rand_like(d) < 0.007) if sparse else None
        cot = torch.randn(B, 1, H, W, device=dev)
        sets.append((g, d, sp, cot))
    m = pkg.CSPN_new.AffinityPropagate(24, 3)
    def run(k):
        g, d, sp, cot = sets[k]
        g = g.clone().requires_grad_(True); d = d.clone().requires_grad_(True)
        m(g, d, sp).backward(cot)
        return g.grad, d.grad
    F.set_resident("off"); refs = [run(k) for k in range(3)]; F.set_resident("on")
    for it in range(400):
        k = it % 3
        gg, gd = run(k)
        n += 1
        if not (torch.equal(gg, refs[k][0]) and torch.equal(gd, refs[k][1])):
            bad += 1
    F.ensure_resident_ok()
    print((B, H, W, sparse), "steps so far", n, "mismatching", bad, flush=True)
print("training steps %d, mismatching %d, fallbacks %d" % (n, bad, F.resident_fallbacks()))
PY
tail -5 $O/train_stress.txt
