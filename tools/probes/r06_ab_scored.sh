#!/bin/bash
# same-box A/B of library variants under _ab/ (CSPN_HIP_LIB) on the scored forwards: config 2 / KITTI B=8 / shards, plain and sparse; three
# rounds interleaved; every line carries a checksum of the refined depth's BITS per shape (variants that claim bit-identity must agree)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for v in "$@"; do
CSPN_HIP_LIB=$PWD/_ab/lib_$v.so python - <<PY
import sys, torch
sys.path.insert(0, ".")
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import evaluation as ev
torch.manual_seed(0)
def clock(fn, n=200):
    for _ in range(30): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 1000 / n)
    return best
new = pkg.CSPN_new.AffinityPropagate(24, 3); acc = ev.new_accumulator("cuda")
out = []
for (B, H, W) in [(24, 228, 304), (8, 352, 1216), (3, 228, 304), (1, 352, 1216)]:
    g12 = torch.randn(B, 12, H, W, device="cuda"); d = torch.rand(B, 1, H, W, device="cuda") * 10; tg = d + 0.1
    sp = torch.where(torch.rand_like(d) < 0.02, tg, torch.zeros_like(d))
    with torch.no_grad():
        a = clock(lambda: new.forward_scored(g12, d, None, tg, acc))
        b = clock(lambda: new.forward_scored(g12, d, sp, tg, acc))
        ck = int(new(g12, d, None).view(torch.int32).long().sum()) ^ int(new(g12, d, sp).view(torch.int32).long().sum())
    out.append("%dx%dx%d %.2f / sparse %.2f [%x]" % (B, H, W, a, b, ck & 0xffffffff))
print("round $r variant $v: " + "   ".join(out) + "  us per scored forward")
PY
done; done
