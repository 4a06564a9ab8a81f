"""Developer probe (round 5): host issue time vs wall time of the scored forward at the per-GPU shard sizes (are the shards host-bound?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import evaluation as ev
torch.manual_seed(0)
new = pkg.CSPN_new.AffinityPropagate(24, 3); acc = ev.new_accumulator("cuda")
for (B, H, W) in [(3, 228, 304), (1, 352, 1216), (24, 228, 304)]:
    g = torch.randn(B, 12, H, W, device="cuda"); d = torch.rand(B, 1, H, W, device="cuda") * 10; tg = d + 0.1
    with torch.no_grad():
        for _ in range(50):
            new.forward_scored(g, d, None, tg, acc)
        torch.cuda.synchronize()
        for rep in range(3):
            n = 500
            t0 = time.perf_counter()
            for _ in range(n):
                new.forward_scored(g, d, None, tg, acc)
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print("%dx%dx%d scored forward: host issue %.2f us, wall %.2f us per call" % (B, H, W, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
        # host-only cost: how long does the host need when the GPU is never the bottleneck?  (issue 50, drain, repeat)
        tot = 0.0
        for _ in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8):
                new.forward_scored(g, d, None, tg, acc)
            tot += time.perf_counter() - t0
        print("   host time with an empty queue: %.2f us per call" % (tot / 160 * 1e6), flush=True)
