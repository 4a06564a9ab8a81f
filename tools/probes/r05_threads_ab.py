"""Developer probe (round 5): the 3x3 resident launch with 1024-thread workgroups at ONE quad per thread (four
wavefronts per SIMD: the LDS write -> barrier -> LDS read round trip of a step hides behind the other wavefronts' FMAs) against the
512-thread workgroups, at per-GPU shard sizes.  Same process, alternating, scored forwards; outputs compared bit for bit."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cspn_monodepth_amd import functional as F, evaluation as ev
torch.manual_seed(0)
acc = ev.new_accumulator("cuda")
res = {}


def full_plan(B, H, W, T, blend, threads):
    """a complete ctypes plan (the C side then skips its tiling search: with a partial plan the call is host-bound on it)"""
    from cspn_monodepth_amd import _lib
    rp = F.resident_plan(B, H, W, T, blend, 0, 0, threads)
    cp = _lib.cspn_resident_plan()
    for name, _ in _lib.cspn_resident_plan._fields_:
        if name != "debug_stamps":
            setattr(cp, name, rp[name])
    return cp


def clock(fn, n=200):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / n)
    return best


for (B, H, W) in [(1, 352, 1216), (2, 352, 1216), (3, 228, 304), (6, 228, 304), (1, 228, 304)]:
    g = torch.randn(B, 12, H, W, device="cuda"); d = torch.rand(B, H, W, device="cuda") * 10; tg = d + 0.1
    sp = torch.where(torch.rand_like(d) < 0.02, tg, torch.zeros_like(d))
    for sparse in (None, sp):
        rows = {}
        outs = {}
        plans = {th: full_plan(B, H, W, 24, int(sparse is not None), th) for th in (512, 1024)}
        with torch.no_grad():
            for rep in range(3):
                for th in (512, 1024):
                    t = clock(lambda: F.forward_resident(g, d, sparse, 24, int(sparse is not None), score=(tg, acc), guard=0, _plan=plans[th]))
                    rows.setdefault(th, []).append(round(t, 2))
            for th in (512, 1024):
                outs[th] = F.forward_resident(g, d, sparse, 24, int(sparse is not None), guard=0, _plan=plans[th])
        torch.cuda.synchronize()
        key = "%dx%dx%d%s" % (B, H, W, "_sparse" if sparse is not None else "")
        res[key] = dict(us_512=rows[512], us_1024=rows[1024], bit_identical=bool(torch.equal(outs[512], outs[1024])),
                        plan_512={k: v for k, v in F.resident_plan(B, H, W, 24, int(sparse is not None), 0, 0, 512).items() if k in ("tiles_x", "tiles_y", "quads_per_thread", "steps_per_phase")},
                        plan_1024={k: v for k, v in F.resident_plan(B, H, W, 24, int(sparse is not None), 0, 0, 1024).items() if k in ("tiles_x", "tiles_y", "quads_per_thread", "steps_per_phase")})
        print(key, res[key], flush=True)
F.ensure_resident_ok()
os.makedirs(os.path.join(ROOT, "gpurun_out", "r05_ab"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r05_ab", "threads_ab.json"), "w"), indent=1)
