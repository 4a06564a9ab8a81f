#!/bin/bash
# A/B of the K = 5 reverse sweep: gather from the forward volume (cspn_propagate_transposed) vs transposed copy + streaming
cd ${GRAFT_REPO_ROOT:-/root/repo}
for dt in f16 f32; do for m in gather copy; do
CSPN_REVERSE_SWEEP=$m python - <<PY
import torch, time, sys
sys.path.insert(0, ".")
import cspn_monodepth_amd as pkg
dt = torch.float16 if "$dt" == "f16" else torch.float32
g = torch.randn(24, 24, 228, 304, device="cuda").to(dt).requires_grad_(True)
d = (torch.rand(24, 1, 228, 304, device="cuda") * 10).to(dt).requires_grad_(True)
cot = torch.randn(24, 1, 228, 304, device="cuda").to(dt)
m = pkg.CSPN_ours.AffinityPropagate(12)
def it():
    g.grad = None; d.grad = None
    m(d, g).backward(cot)
for _ in range(5): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): it()
torch.cuda.synchronize(); print("$dt $m: %.1f us per fwd+bwd" % ((time.perf_counter() - t0) / 30 * 1e6))
PY
done; done
