"""Developer probe: where does the HOST time of a forward + backward of the module go?  cProfile over a few hundred steps
(the GPU queue is never waited for inside the loop), next to the GPU time per step from HIP events."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cspn_monodepth_amd as pkg

dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "pac5"
B, H, W = 24, 228, 304
if which == "pac5":
    dt, K, T = torch.float16, 5, 12
    g = torch.randn(B, 24, H, W, device=dev).to(dt).requires_grad_(True)
    m = pkg.CSPN_ours.AffinityPropagate(T)
else:
    dt, K, T = torch.float32, 3, 24
    g = torch.randn(B, 12, H, W, device=dev).to(dt).requires_grad_(True)
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
d = (torch.rand(B, 1, H, W, device=dev) * 10).to(dt).requires_grad_(True)
cot = torch.randn(B, 1, H, W, device=dev).to(dt)


def step():
    g.grad = None
    d.grad = None
    out = m(d, g, None) if which == "pac5" else m(g, d, None)
    out.backward(cot)


for _ in range(20):
    step()
torch.cuda.synchronize()
N = 300
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(N):
    step()
t_host = time.perf_counter() - t0
e1.record()
torch.cuda.synchronize()
print("%s: host issue time %.1f us/step, GPU time %.1f us/step" % (which, t_host / N * 1e6, e0.elapsed_time(e1) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:50]))
