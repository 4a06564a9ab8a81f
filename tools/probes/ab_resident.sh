#!/bin/bash
# same-box A/B of libcspn_hip.so variants under _ab/ (CSPN_HIP_LIB): bench config 2, three rounds interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for v in "$@"; do
CSPN_HIP_LIB=$PWD/_ab/lib_$v.so python - <<PY
import sys, torch, time
sys.path.insert(0, ".")
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F, evaluation as ev
g = torch.randn(24, 12, 228, 304, device="cuda"); d = torch.rand(24, 1, 228, 304, device="cuda") * 10
tg = d + 0.1
m = pkg.CSPN_new.AffinityPropagate(24, 3); acc = ev.new_accumulator("cuda")
with torch.no_grad():
    for _ in range(50): m.forward_scored(g, d, None, tg, acc)
    torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): m.forward_scored(g, d, None, tg, acc)
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 10)
print("round $r variant $v: %.2f us per scored forward" % best)
PY
done; done
