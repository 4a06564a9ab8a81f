#!/usr/bin/env python3
"""Print the north-star table from the bench lines tools/scale_sweep.sh wrote (scale_<workload>_n<N>.json): per workload and N
the whole-job depth-maps/s, the scaling efficiency against N x the N = 1 line (weak scaling: nyu / pac5 / train; kitti shards ONE
batch of 8: strong), and the fraction of N x the HBM roofline of the workload's algorithmic bytes ((K*K+1) *
sizeof(dtype) bytes per pixel and step at 8 TB/s per GPU; `train` has no such roofline — the CSPN module is 0.3 % of that step)."""
import glob
import json
import os
import re
import sys

HBM = 8.0e12
ALGO = {"nyu": 24 * 69312 * 40.0, "kitti": 24 * 428032 * 40.0, "pac5": 12 * 69312 * 52.0}      # bytes per depth map


def main(d):
    rows = {}
    for p in sorted(glob.glob(os.path.join(d, "scale_*_n*.json"))):
        m = re.match(r"scale_(\w+)_n(\d+)\.json", os.path.basename(p))
        try:
            line = json.load(open(p))
        except (ValueError, OSError):
            continue
        rows.setdefault(m.group(1), {})[int(m.group(2))] = line
    if not rows:
        print("no bench lines under", d)
        return 1
    print("%-6s %3s %14s %10s %12s %10s  %s" % ("wl", "N", "depth-maps/s", "ms/step", "efficiency", "N x HBM", "scaling / parallelism"))
    for wl, by_n in rows.items():
        base = by_n.get(1)
        for n in sorted(by_n):
            ln = by_n[n]
            eff = ln["value"] / (n * base["value"]) if base else float("nan")
            roof = ln["value"] * ALGO[wl] / (n * HBM) if wl in ALGO else float("nan")
            print("%-6s %3d %14.0f %10.4f %11.1f%% %9.1f%%  %s / %s" % (
                wl, n, ln["value"], ln["ms_per_step"], 100 * eff, 100 * roof, ln.get("scaling"), ln.get("config", {}).get("parallelism")))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/scale"))
