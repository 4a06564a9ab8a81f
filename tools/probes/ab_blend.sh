#!/bin/bash
# same-box A/B of libcspn_hip.so variants under _ab/ (CSPN_HIP_LIB): the BLENDED resident forms — CSPN_new inference,
# CSPN_ours K = 3 fp32 inference (the reference model's call), CSPN_new training forward + backward; three rounds interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for v in "$@"; do
CSPN_HIP_LIB=$PWD/_ab/lib_$v.so python - <<PY
import sys, torch
sys.path.insert(0, ".")
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import evaluation as ev
torch.manual_seed(0)
B, H, W = 24, 228, 304
g12 = torch.randn(B, 12, H, W, device="cuda"); g8 = torch.randn(B, 8, H, W, device="cuda")
d = torch.rand(B, 1, H, W, device="cuda") * 10; tg = d + 0.1
sp = torch.where(torch.rand_like(d) < 0.02, tg, torch.zeros_like(d))
new = pkg.CSPN_new.AffinityPropagate(24, 3); ours = pkg.CSPN_ours.AffinityPropagate(24); acc = ev.new_accumulator("cuda")
def clock(fn, n=100):
    for _ in range(30): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 1000 / n)
    return best
with torch.no_grad():
    a = clock(lambda: new.forward_scored(g12, d, sp, tg, acc))
    b = clock(lambda: ours.forward_scored(d, g8, sp, tg, acc))
    c = clock(lambda: new.forward_scored(g12, d, None, tg, acc))
gt = g12.clone().requires_grad_(True); dt = d.clone().requires_grad_(True)
def train():
    out = new(gt, dt, sp); out.backward(tg); gt.grad = None; dt.grad = None
t = clock(train, 50)
print("round $r variant $v: new+sparse %.2f  ours K3 fp32+sparse %.2f  new no-sparse %.2f  train new+sparse %.1f us" % (a, b, c, t))
PY
done; done
