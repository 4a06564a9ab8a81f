"""Developer probe (round 5): unprofiled HOST time of the pieces of a 3x3 training-shaped step — thin perf_counter wrappers around
the functions of functional.py that a forward + backward passes through (cProfile cannot see the autograd engine's worker thread,
where the backward runs)."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F, _lib
DEV = "cuda:0"
B, H, W, T = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (24, 228, 304, 24)))
torch.manual_seed(0)
g = torch.randn(B, 12, H, W, device=DEV); d = torch.rand(B, 1, H, W, device=DEV) * 10
gt = g.clone().requires_grad_(True); dt = d.clone().requires_grad_(True); cot = torch.randn_like(d)
m = pkg.CSPN_new.AffinityPropagate(T, 3)
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
    out = m(gt, dt, None)
    t1 = time.perf_counter()
    out.backward(cot)
    t2 = time.perf_counter()
    acc["TOTAL forward (module call)"] += t1 - t0; acc["TOTAL backward (.backward())"] += t2 - t1


for _ in range(30):
    fwd_bwd()
torch.cuda.synchronize()
acc.clear(); cnt.clear()
for name in ("forward_resident", "_resident_launch", "_journal_add", "resident_supported", "from_guidance_supported", "_reverse_sweep",
             "transposed_resident_guidance", "_check_resident_at_end_of_backward", "_plane", "_resident_plan_cached", "_with_spin_limit"):
    wrap(F, name)
wrap(F.CSPN3Function, "forward", "CSPN3Function.forward"); wrap(F.CSPN3Function, "backward", "CSPN3Function.backward")
L = _lib.lib()
for name in ("cspn3_forward_resident", "cspn3_transposed_resident_guidance", "cspn3_backward_tail"):
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
