"""Developer probe: how much of the host time of a forward + backward step of the 3x3 module is OUR Python (CSPN3Function.forward /
.backward bodies) and how much is PyTorch's autograd machinery around it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
dev = "cuda:0"
B, H, W, T = 24, 228, 304, 24
g = torch.randn(B, 12, H, W, device=dev).requires_grad_(True)
d = (torch.rand(B, 1, H, W, device=dev) * 10).requires_grad_(True)
cot = torch.randn(B, 1, H, W, device=dev)
m = pkg.CSPN_new.AffinityPropagate(T, 3)
acc = {"fwd": 0.0, "bwd": 0.0}
f0, b0 = F.CSPN3Function.forward, F.CSPN3Function.backward
def fwd(ctx, *a):
    t = time.perf_counter(); r = f0(ctx, *a); acc["fwd"] += time.perf_counter() - t; return r
def bwd(ctx, *a):
    t = time.perf_counter(); r = b0(ctx, *a); acc["bwd"] += time.perf_counter() - t; return r
F.CSPN3Function.forward = staticmethod(fwd)
F.CSPN3Function.backward = staticmethod(bwd)
def step():
    g.grad = None; d.grad = None
    m(g, d, None).backward(cot)
for _ in range(20): step()
torch.cuda.synchronize()
for k in acc: acc[k] = 0.0
N = 300
t0 = time.perf_counter()
for _ in range(N): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("check %s: host %.1f us/step (wall %.1f), of which CSPN3Function.forward body %.1f, .backward body %.1f, autograd + module + loop %.1f" % (
    os.environ.get("CSPN_BWD_CHECK", "on"), th / N * 1e6, tt / N * 1e6, acc["fwd"] / N * 1e6, acc["bwd"] / N * 1e6, (th - acc["fwd"] - acc["bwd"]) / N * 1e6))

# ---- finer split of the bodies (perf_counter around the package's own helpers)
import collections
F.CSPN3Function.forward = staticmethod(f0); F.CSPN3Function.backward = staticmethod(b0)
split = collections.OrderedDict()
def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); split[name] = split.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(mod, name, w)
for n in ("forward_resident", "transposed_resident", "_reverse_sweep", "_check_resident_at_end_of_backward", "_resident_launch", "resident_supported", "_plane", "_tail_vector_ok"):
    wrap(F, n)
L = F._lib.lib()
tail0 = L.cspn3_backward_tail
def tail(*a):
    t = time.perf_counter(); r = tail0(*a); split["C cspn3_backward_tail"] = split.get("C cspn3_backward_tail", 0.0) + time.perf_counter() - t; return r
class LW(object):
    def __getattr__(self, k): return tail if k == "cspn3_backward_tail" else getattr(L, k)
lw = LW()
F._lib.lib = lambda: lw
emp0 = torch.empty
def emp(*a, **k):
    t = time.perf_counter(); r = emp0(*a, **k); split["torch.empty"] = split.get("torch.empty", 0.0) + time.perf_counter() - t; return r
torch.empty = emp
for _ in range(10): step()
torch.cuda.synchronize(); split.clear()
for _ in range(N): step()
torch.cuda.synchronize()
print({k: round(v / N * 1e6, 1) for k, v in split.items()})
