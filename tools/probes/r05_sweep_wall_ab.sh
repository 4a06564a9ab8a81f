#!/bin/bash
# wall time of the K = 5 fp16 training step (config 3's shape) per library variant under _ab/, N interleaved processes each (the step is
# placement-sensitive with the host threads unbound: read the distribution, not one number).  usage: r05_sweep_wall_ab.sh <rounds> <tags...>
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for r in $(seq $N); do for v in "$@"; do
CSPN_HIP_LIB=$PWD/_ab/lib_$v.so python - <<PY 2>/dev/null
import sys, torch
sys.path.insert(0, ".")
import cspn_monodepth_amd as pkg
torch.manual_seed(0)
B, H, W, T = 24, 228, 304, 12
g = torch.randn(B, 24, H, W, device="cuda").half().requires_grad_(True)
d = (torch.rand(B, 1, H, W, device="cuda") * 10).half().requires_grad_(True)
cot = torch.randn(B, 1, H, W, device="cuda").half()
m = pkg.CSPN_ours.AffinityPropagate(T, state_dtype=None)
def step():
    g.grad = None; d.grad = None
    m(d, g).backward(cot)
for _ in range(20): step()
torch.cuda.synchronize(); ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): step()
    e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1000 / 50)
print("$v #$r: min %.1f median %.1f max %.1f us" % (min(ts), sorted(ts)[3], max(ts)))
PY
done; done
