#!/usr/bin/env python3
"""Brute-force sweep of launch plans for the propagation loop on the GPU box (developer tool).

    python tools/tune.py --workload nyu --out gpurun_out/tune_nyu.jsonl

Times cspn_propagate (T steps, no prepare) for every plan that fits, with HIP events, and prints the best.
Also times the prepare and metrics kernels and the full module forward.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cspn_monodepth_amd as pkg                      # noqa: E402
from cspn_monodepth_amd import functional as F        # noqa: E402
from bench import WORKLOADS, make_inputs              # noqa: E402


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="nyu")
    ap.add_argument("--out", default="")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--sparse", action="store_true")
    ap.add_argument("--S", default="1,2,3,4,5,6,7,8,9,10,12")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--from-guidance", action="store_true", help="time the fused prepare+propagate entry (K=3)")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    dev = torch.device("cuda", 0)
    B, H, W, K, T = wl["B"], wl["H"], wl["W"], wl["K"], wl["T"]
    g, d, s, target = make_inputs(wl, B, dev, 1, args.sparse)
    R = K // 2
    with torch.no_grad():
        if K == 3:
            w, _, _ = F.cspn3_prepare(g)
            t_prep = timed(lambda: F.cspn3_prepare(g), 20)
        else:
            w, _ = F.pac_prepare(g)
            t_prep = timed(lambda: F.pac_prepare(g), 20)
        d0 = d[:, 0].contiguous()
        sp = None if s is None else s[:, 0].contiguous()
        blend = F.BLEND_SPARSE if sp is not None else F.BLEND_NONE
        ref, _ = F.propagate(w, d0, sp, K, T, blend, plan=dict(steps_per_launch=1, tile_w=64, tile_h=16,
                                                               quads_per_thread=1, threads=256))
        out = torch.empty_like(ref)
        acc = pkg.evaluation.new_accumulator(dev)
        tgt = target[:, 0].contiguous()
        t_met = timed(lambda: pkg.evaluation.metric_sums(ref, tgt, out=acc), 20)
        print("prepare %.1f us   metrics %.1f us" % (t_prep, t_met), flush=True)
        esz = 2 if wl["dtype"] == "f16" else 4
        alg = (K * K + 1 + (2 if sp is not None else 0)) * esz * B * H * W * T
        rows = []
        nqs = {3: (1, 2, 4, 8), 5: (1, 2, 3), 7: (1,)}[K]
        t_start = time.time()
        for S in [int(x) for x in args.S.split(",")]:
            if S > T:
                continue
            hyw = (S - 1) * R
            hxw = (hyw + 3) // 4 * 4
            for threads in (256, 512, 1024):
                for nq in nqs:
                    for tw in sorted(set(range(16, 161, 8)) | {(-(-W // n) + 3) // 4 * 4 for n in range(1, 17)}):
                        if tw > W + 3:
                            continue
                        wq = (tw + 2 * hxw) // 4
                        if wq > threads:
                            continue
                        th_max = nq * (threads // wq) - 2 * hyw
                        if th_max < 4:
                            continue
                        ths = {th_max, -(-H // -(-H // th_max))}
                        for th in sorted(ths):
                            plan = dict(steps_per_launch=S, tile_w=tw, tile_h=th, quads_per_thread=nq, threads=threads)
                            try:
                                F.resolve_plan(K, B, H, W, T, False, plan)
                                if args.from_guidance:
                                    us = timed(lambda: F.propagate_from_guidance(g, d0, sp, T, blend, plan=plan), args.reps, 1)
                                else:
                                    us = timed(lambda: F.propagate(w, d0, sp, K, T, blend, plan=plan), args.reps, 1)
                            except RuntimeError:
                                continue
                            rows.append(dict(plan, us=us, alg_GBs=alg / us / 1e3))
        rows.sort(key=lambda r: r["us"])
        print("swept %d plans in %.1f s" % (len(rows), time.time() - t_start))
        best_by_s = {}
        for r in rows:
            best_by_s.setdefault(r["steps_per_launch"], r)
        for S in sorted(best_by_s):
            r = best_by_s[S]
            o2, _ = F.propagate(w, d0, sp, K, T, blend, plan={k: r[k] for k in ("steps_per_launch", "tile_w", "tile_h", "quads_per_thread", "threads")})
            err = float(((o2.float() - ref.float()).abs() / ref.float().abs().clamp_min(1e-6)).max())
            print("S=%2d best: tile %3dx%-3d nq=%d thr=%d  %.1f us / %d steps = %.2f us/step  alg %.0f GB/s  (rel diff vs S=1: %.1e)" % (
                S, r["tile_w"], r["tile_h"], r["quads_per_thread"], r["threads"], r["us"], T, r["us"] / T, r["alg_GBs"], err))
        for r in rows[:12]:
            print(r)
        # full module forward with the best plan and with the default
        if K == 3:
            for name, plan in (("default", None), ("best", {k: rows[0][k] for k in ("steps_per_launch", "tile_w", "tile_h", "quads_per_thread", "threads")})):
                m = pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)
                us = timed(lambda: m(g, d, s), 20)
                print("module forward (%s plan): %.1f us -> %.0f maps/s" % (name, us, B / us * 1e6))
        if args.out:
            with open(args.out, "w") as f:
                for r in rows:
                    f.write(json.dumps(r) + "\n")
        del out


if __name__ == "__main__":
    main()
