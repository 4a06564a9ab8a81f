#!/usr/bin/env python3
"""Developer probe for cspnk_resident: (1) where does it differ from the multi-launch schedule, (2) per-phase timeline from
the in-kernel stamps, (3) event timings over T (derive vs steps vs exchanges)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                         # noqa: E402
import cspn_monodepth_amd as pkg                     # noqa: E402
from cspn_monodepth_amd import functional as F       # noqa: E402

dev = "cuda:0"
K, B, H, W, S = 5, int(os.environ.get("PB", 24)), 228, 304, int(os.environ.get("PS", 4))
torch.manual_seed(0)
g = torch.randn(B, K * K - 1, H, W, device=dev).half()
x = (torch.rand(B, 1, H, W, device=dev) * 10).half()


def multi(T, S):
    prev = F._RESIDENT_MODE
    F.set_resident("off")
    try:
        with torch.no_grad():
            return pkg.CSPN_ours.AffinityPropagate(T, plan=dict(steps_per_launch=S), state_dtype=None)(x, g)[:, 0]
    finally:
        F.set_resident(prev)


if "diff" in sys.argv:
    rp = F.kres_plan(K, B, H, W, 12, 0, 0, S)
    print("plan", rp)
    for T in (1, 2, 3, 4, 5, 8, 12):
        with torch.no_grad():
            out = F.pac_forward_resident(g, x[:, 0].contiguous(), None, T, steps_per_phase=S)
        ref = multi(T, S)
        torch.cuda.synchronize()
        bad = (out != ref)
        n = int(bad.sum())
        msg = "T=%d mismatches %d / %d" % (T, n, out.numel())
        if n:
            idx = bad.nonzero()[:2000].cpu()
            d = (out.float() - ref.float()).abs()
            ys, xs = idx[:, 1], idx[:, 2]
            msg += " max %.4g; y%%th hist %s; x%%tw hist %s; first %s" % (
                float(d.max()), torch.bincount(ys % rp["tile_h"], minlength=rp["tile_h"]).tolist(),
                torch.bincount((xs % rp["tile_w"]) // 8, minlength=rp["tile_w"] // 8).tolist(), idx[:6].tolist())
        print(msg)
if "weights" in sys.argv:
    # x = 1 on one residue class of (y % 5, x % 5): every output then equals ONE tap weight (or 0): a direct comparison of the
    # softmax weights of the two schedules
    total = 0
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    x_keep = x
    for a in range(5):
        for b_ in range(5):
            x = ((yy % 5 == a) & (xx % 5 == b_)).half().expand(B, 1, H, W).contiguous()
            with torch.no_grad():
                out = F.pac_forward_resident(g, x[:, 0].contiguous(), None, 1, steps_per_phase=1)
            ref = multi(1, 1)
            n = int((out != ref).sum())
            total += n
            if n:
                idx = (out != ref).nonzero()[:3].cpu().tolist()
                print("pattern", a, b_, "mismatching weights", n, idx, [(float(out[tuple(i)]), float(ref[tuple(i)])) for i in idx])
    print("weight mismatches in all 25 patterns:", total)
    x = x_keep
    # all-ones state: out = half(sum of the weights accumulated in tap order)
    x = torch.ones_like(x)
    with torch.no_grad():
        out = F.pac_forward_resident(g, x[:, 0].contiguous(), None, 1, steps_per_phase=1)
    ref = multi(1, 1)
    print("all-ones state: mismatches", int((out != ref).sum()))
    x = x_keep
if "stamps" in sys.argv:
    for T, SS in ((12, 4), (12, 6), (4, 4)):
        TH = int(os.environ.get("PT", 0))
        rp = F.kres_plan(K, B, H, W, T, 0, 0, SS, TH)
        grid = rp["tiles_x"] * rp["tiles_y"] * min(B, rp["images_per_launch"])
        st = torch.zeros((grid, 16), dtype=torch.int64, device=dev)
        names = ["parked", "derive"]
        for p in range(-(-T // rp["steps_per_phase"])):
            names += ["stage%d" % p, "steps%d" % p, "xchg%d" % p]
        names[-1] = "epilogue"
        with torch.no_grad():
            for _ in range(3):
                F.pac_forward_resident(g, x[:, 0].contiguous(), None, T, steps_per_phase=SS, debug_stamps=st, threads=TH)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            F.pac_forward_resident(g, x[:, 0].contiguous(), None, T, steps_per_phase=SS, debug_stamps=st, threads=TH)
            e1.record()
            e1.synchronize()
        t = st.cpu().numpy().astype("float64") / 100.0
        t0 = t[:, 0].min()
        print("T=%d S=%d plan %s; call (events) %.1f us; stamps are of the LAST launch (%d workgroups)" % (
            T, SS, {k: rp[k] for k in ("tiles_x", "tiles_y", "tile_w", "tile_h", "quads_per_thread", "threads", "images_per_launch", "launches")},
            e0.elapsed_time(e1) * 1e3, grid))
        print("  workgroup start spread %.2f us" % (t[:, 0].max() - t0))
        for k in range(1, 16):
            if t[:, k].max() == 0 or k > len(names):
                break
            dt = t[:, k] - t[:, k - 1]
            print("  %-9s mean %.2f  min %.2f  max %.2f   (ends %.2f .. %.2f us)" % (names[k - 1], dt.mean(), dt.min(), dt.max(),
                                                                                 t[:, k].min() - t0, t[:, k].max() - t0))
if "plans" in sys.argv:
    xq = x[:, 0].contiguous()
    tg = (xq.float() + 0.1).half()
    from cspn_monodepth_amd import evaluation as ev
    acc = ev.new_accumulator(dev)
    for SS in (2, 4, 6):
        for th in (512, 768):
            rp = F.kres_plan(K, B, H, W, 12, 0, 0, SS, th)
            if rp is None:
                print("S=%d threads=%d: no plan" % (SS, th))
                continue
            res = []
            for score in (None, (tg, acc)):
                with torch.no_grad():
                    for _ in range(3):
                        F.pac_forward_resident(g, xq, None, 12, steps_per_phase=SS, threads=th, score=score)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        F.pac_forward_resident(g, xq, None, 12, steps_per_phase=SS, threads=th, score=score)
                    e1.record()
                    e1.synchronize()
                res.append(e0.elapsed_time(e1) * 1e3 / 20)
            print("S=%d threads=%d: %.1f us plain, %.1f us scored; %s" % (SS, th, res[0], res[1], {k: rp[k] for k in (
                "tiles_x", "tiles_y", "tile_w", "tile_h", "quads_per_thread", "images_per_launch", "launches", "region_over_tile")}))
if "f32" in sys.argv:
    # the reference model's own configuration: K = 3, 8-channel fp32 guidance, T = 24 (unet_ours.py:279, :305, :333)
    for Kq, Tq in ((3, 24), (5, 12)):
        gq = torch.randn(B, Kq * Kq - 1, H, W, device=dev)
        xq = torch.rand(B, 1, H, W, device=dev) * 10
        mq = pkg.CSPN_ours.AffinityPropagate(Tq)

        def timeit(fn):
            with torch.no_grad():
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                e1.synchronize()
            return e0.elapsed_time(e1) * 1e3 / 20
        F.set_resident("off")
        t_multi = timeit(lambda: mq(xq, gq))
        F.set_resident("auto")
        print("fp32 K=%d T=%d B=%d: multi-launch %.1f us" % (Kq, Tq, B, t_multi))
        for SS in (0, 4, 6, 8):
            for th in (0, 512, 768):
                rp = F.kres_plan(Kq, B, H, W, Tq, 0, 0, SS, th, F.CSPN_F32)
                if rp is None:
                    continue
                t = timeit(lambda: F.pac_forward_resident(gq, xq[:, 0].contiguous(), None, Tq, steps_per_phase=SS, threads=th))
                print("   resident S=%d threads=%d: %.1f us  %s" % (SS, th, t, {k: rp[k] for k in (
                    "steps_per_phase", "tiles_x", "tiles_y", "quads_per_thread", "threads", "images_per_launch", "launches")}))
if "sweep" in sys.argv:
    for Bq in (12, 24):
        gq, xq = g[:Bq].contiguous(), x[:Bq, 0].contiguous()
        for T, SS in ((2, 2), (4, 4), (8, 4), (12, 4), (6, 6), (12, 6), (12, 2)):
            with torch.no_grad():
                for _ in range(3):
                    F.pac_forward_resident(gq, xq, None, T, steps_per_phase=SS)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    F.pac_forward_resident(gq, xq, None, T, steps_per_phase=SS)
                e1.record()
                e1.synchronize()
            print("B=%d T=%d S=%d: %.1f us per call" % (Bq, T, SS, e0.elapsed_time(e1) * 1e3 / 20))
F.ensure_resident_ok()
