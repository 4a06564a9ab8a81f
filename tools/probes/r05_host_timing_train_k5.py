"""Developer probe (round 5): unprofiled HOST time of the pieces of the K = 5 fp16 training-shaped step (config 3's shape) — thin perf_counter wrappers around
the functions of functional.py that a forward + backward passes through (cProfile cannot see the autograd engine's worker thread,
where the backward runs)."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F, _lib
DEV = "cuda:0"
B, H, W, T = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (24, 228, 304, 12)))
torch.manual_seed(0)
g = torch.randn(B, 24, H, W, device=DEV).half(); d = (torch.rand(B, 1, H, W, device=DEV) * 10).half()
gt = g.clone().requires_grad_(True); dt = d.clone().requires_grad_(True); cot = torch.randn(B, 1, H, W, device=DEV).half()
m = pkg.CSPN_ours.AffinityPropagate(T, state_dtype=None)
acc = collections.defaultdict(float); cnt = collections.Counter()


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0; cnt[label] += 1
    setattr(obj, name, w)


def fwd_bwd():
    gt.grad = None; dt.grad = None
    t0 = time.perf_counter()
    out = m(dt, gt)
    t1 = time.perf_counter()
    out.backward(cot)
    t2 = time.perf_counter()
    acc["TOTAL forward (module call)"] += t1 - t0; acc["TOTAL backward (.backward())"] += t2 - t1


for _ in range(30):
    fwd_bwd()
torch.cuda.synchronize()
acc.clear(); cnt.clear()
for name in ("pac_forward_resident_history", "pac_resident_supported", "pac_transposed_resident", "_kres_plan_cached", "_resident_launch", "_journal_add", "resident_supported", "from_guidance_supported", "_reverse_sweep",
             "transposed_resident_guidance", "_check_resident_at_end_of_backward", "_plane", "_resident_plan_cached", "_with_spin_limit"):
    wrap(F, name)
wrap(F.PACFunction, "forward", "PACFunction.forward"); wrap(F.PACFunction, "backward", "PACFunction.backward")
L = _lib.lib()
for name in ("cspnk_forward_resident_history", "cspnk_transposed_resident", "cspn_pac_backward_tail"):
    wrap(L, name, "C:" + name)
wrap(torch, "empty", "torch.empty"); wrap(torch, "empty_like", "torch.empty_like")
N = 300
t0 = time.perf_counter()
for _ in range(N):
    fwd_bwd()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("shape %dx%dx%d T=%d: host issue %.1f us per pass, wall %.1f us per pass (with the wrappers)" % (B, H, W, T, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-45s %7.1f us per pass  (%d calls)" % (k, v / N * 1e6, cnt[k] // N if cnt[k] else 1))
