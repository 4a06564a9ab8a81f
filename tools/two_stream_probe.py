#!/usr/bin/env python3
"""Does splitting the batch over two HIP streams (independent dependent-launch chains) fill the tail of each launch?
Both variants are captured into a HIP graph so that host time does not matter (developer tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_monodepth_amd as pkg
from tools.tune import timed
from bench import WORKLOADS, make_inputs
wl = dict(WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "nyu"])
B = wl["B"]
g, d, s, tgt = make_inputs(wl, B, torch.device("cuda", 0), 1, False)
m = pkg.CSPN_new.AffinityPropagate(wl["T"], 3)
acc = pkg.evaluation.new_accumulator(g.device)


def capture(nsplit):
    bounds = [pkg.evaluation.shard_bounds(B, r, nsplit) for r in range(nsplit)]
    parts = [(g[lo:hi], d[lo:hi], tgt[lo:hi]) for lo, hi in bounds]
    side = [torch.cuda.Stream() for _ in range(nsplit - 1)]

    def run():
        cur = torch.cuda.current_stream()
        outs = []
        for i, (gg, dd, tt) in enumerate(parts):
            if i == 0:
                outs.append(m.forward_scored(gg, dd, None, tt, acc))
            else:
                st = side[i - 1]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs.append(m.forward_scored(gg, dd, None, tt, acc))
        for st in side:
            cur.wait_stream(st)
        return outs
    with torch.no_grad():
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(3):
                run()
        torch.cuda.current_stream().wait_stream(warm)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = run()
    return graph, outs


ref = None
for nsplit in (1, 2, 3, 4):
    graph, outs = capture(nsplit)
    us = timed(graph.replay, 200, 20)
    out = torch.cat(outs, 0)
    if ref is None:
        ref = out.clone()
    print("%d stream(s): %.1f us per batch of %d -> %.0f maps/s   (same result: %s)" % (
        nsplit, us, B, B / us * 1e6, bool(torch.equal(out, ref))), flush=True)
