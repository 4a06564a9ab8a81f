#!/bin/bash
# same-box A/B of library variants under _ab/ (CSPN_HIP_LIB) on config 3 (5x5 softmax, 12 steps, fp16): scored forwards at B = 24 and B = 3,
# plain and sparse, with a checksum of the refined depth (the variants must agree bit for bit); three rounds interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for v in "$@"; do
CSPN_HIP_LIB=$PWD/_ab/lib_$v.so python - <<PY
import sys, torch
sys.path.insert(0, ".")
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import evaluation as ev
torch.manual_seed(0)
def clock(fn, n=100):
    for _ in range(20): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 1000 / n)
    return best
m = pkg.CSPN_ours.AffinityPropagate(12, state_dtype=None); acc = ev.new_accumulator("cuda")
out = []
for (B, H, W) in [(24, 228, 304), (3, 228, 304)]:
    gd = torch.randn(B, 24, H, W, device="cuda").half(); x = (torch.rand(B, 1, H, W, device="cuda") * 10).half(); tg = (x.float() + 0.1).half()
    sp = torch.where(torch.rand(B, 1, H, W, device="cuda") < 0.02, tg, torch.zeros_like(tg))
    with torch.no_grad():
        a = clock(lambda: m.forward_scored(x, gd, None, tg, acc))
        b = clock(lambda: m.forward_scored(x, gd, sp, tg, acc))
        chk = float(m(x, gd, sparse_depth=sp).double().sum())
    out.append("%dx%dx%d %.2f / sparse %.2f (sum %.6f)" % (B, H, W, a, b, chk))
print("round $r variant $v: " + "   ".join(out) + "  us per scored forward")
PY
done; done
