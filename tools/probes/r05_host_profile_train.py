"""Developer probe (round 5): where does the HOST time of a 3x3 training-shaped step (forward with history + backward) go?
cProfile over 300 passes at config 2 + wall per pass with the host threads unbound."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
DEV = "cuda:0"
B, H, W, T = 24, 228, 304, 24
torch.manual_seed(0)
g = torch.randn(B, 12, H, W, device=DEV); d = torch.rand(B, 1, H, W, device=DEV) * 10
gt = g.clone().requires_grad_(True); dt = d.clone().requires_grad_(True); cot = torch.randn_like(d)
m = pkg.CSPN_new.AffinityPropagate(T, 3)


def fwd_bwd():
    gt.grad = None; dt.grad = None
    out = m(gt, dt, None)
    out.backward(cot)


for _ in range(20):
    fwd_bwd()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        fwd_bwd()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d: host issue %.1f us per pass, wall %.1f us per pass" % (rep, (t1 - t0) * 1e4, (t2 - t0) * 1e4), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    fwd_bwd()
pr.disable()
torch.cuda.synchronize()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(45)
print(buf.getvalue()[:9000])
