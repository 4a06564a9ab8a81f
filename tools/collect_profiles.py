#!/usr/bin/env python3
"""Copy the summaries of a round's GPU sessions from gpurun_out/ (scratch) into profiles/ (tracked): bench lines, rocprofv3
kernel stats, PMC traffic / SQ summaries, the in-kernel timelines.  `python tools/collect_profiles.py r03 r03s6`"""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, sess = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(P, dst))
        print("profiles/%s" % dst)


for f in glob.glob(os.path.join(G, rnd + "lines", rnd + "_bench_*.json")):
    cp(f, os.path.basename(f))
for wl, tag in (("nyu", "nyu"), ("pac5", "pac5"), ("kitti", "kitti"), ("nyu_sparse", "nyu_sparse")):
    d = os.path.join(G, "%sprof_%s" % (rnd, wl))
    cp(os.path.join(d, "stats_default_cspn.csv"), "%s_bench_%s_kernel_stats.csv" % (rnd, tag))
    cp(os.path.join(d, "stats_s1_cspn.csv"), "%s_kernel_stats_s1%s.csv" % (rnd, "" if tag == "nyu" else "_" + tag))
    cp(os.path.join(d, "stats_multi_cspn.csv"), "%s_kernel_stats_multi_launch%s.csv" % (rnd, "" if tag == "nyu" else "_" + tag))
    cp(os.path.join(d, "stats_bwd_cspn.csv"), "%s_kernel_stats_train_leg%s.csv" % (rnd, "" if tag == "nyu" else "_" + tag))
    cp(os.path.join(d, "traffic_%s.json" % tag), "%s_pmc_traffic_%s.json" % (rnd, tag))
    cp(os.path.join(d, "traffic_%sbwd.json" % tag), "%s_pmc_traffic_%sbwd.json" % (rnd, tag))
    cp(os.path.join(d, "sq_%s.json" % tag), "%s_sq_%s.json" % (rnd, tag))
S = os.path.join(G, sess)
cp(os.path.join(S, "stats_bwd_pac5_cspn.csv"), "%s_kernel_stats_train_leg_pac5.csv" % rnd)
cp(os.path.join(S, "traffic_pac5bwd.json"), "%s_pmc_traffic_pac5bwd.json" % rnd)
cp(os.path.join(S, "kres_probe.txt"), "%s_kres_plans.txt" % rnd)
cp(os.path.join(S, "resident_s_sweep.txt"), "%s_resident_s_sweep.txt" % rnd)
# the in-kernel timelines (tools/resident_stamps.py, tools/probes/kres_probe.py stamps) as one JSON
tl = {}
for name in ("nyu", "nyu_b3", "kitti_b1"):
    f = os.path.join(S, "resident_timeline_%s.txt" % name)
    if not os.path.exists(f):
        continue
    entry = {"phases": []}
    for line in open(f):
        m = re.match(r"plan (\{.*\}) kernel \(events\) ([\d.]+) us", line)
        if m:
            entry["plan"] = {k: v for k, v in eval(m.group(1)).items() if k not in ("debug_stamps", "spin_limit")}   # noqa: S307 (our own tool's repr)
            entry["call_us_with_stamps"] = float(m.group(2))
        m = re.match(r"(\w+)\s+mean ([\d.]+)\s+min ([\d.]+)\s+max ([\d.]+)\s+\(ends at ([\d.]+) \.\. ([\d.]+) us", line)
        if m:
            entry["phases"].append({"phase": m.group(1), "mean_us": float(m.group(2)), "min_us": float(m.group(3)),
                                    "max_us": float(m.group(4)), "ends_us": [float(m.group(5)), float(m.group(6))]})
    tl["cspn3_resident_" + name] = entry
f = os.path.join(S, "kres_probe.txt")
if os.path.exists(f):
    cur = None
    for line in open(f):
        m = re.match(r"T=(\d+) S=(\d+) plan (\{.*\}); call \(events\) ([\d.]+) us", line)
        if m:
            cur = {"plan": eval(m.group(3)), "call_us_with_stamps": float(m.group(4)), "phases": []}   # noqa: S307
            tl["cspnk_d2_pac5_T%s_S%s" % (m.group(1), m.group(2))] = cur
        m = re.match(r"\s+(\w+)\s+mean ([\d.]+)\s+min ([\d.]+)\s+max ([\d.]+)\s+\(ends ([\d.]+) \.\. ([\d.]+) us", line)
        if m and cur is not None:
            cur["phases"].append({"phase": m.group(1), "mean_us": float(m.group(2)), "min_us": float(m.group(3)),
                                  "max_us": float(m.group(4)), "ends_us": [float(m.group(5)), float(m.group(6))]})
if tl:
    tl["_note"] = ("in-kernel 100 MHz wall-clock stamps of thread 0 of every workgroup (tools/resident_stamps.py, tools/probes/"
                   "kres_probe.py stamps); a phase's time is stamp k - stamp k-1 per workgroup; `ends_us` = earliest .. latest "
                   "workgroup relative to the first workgroup's start")
    json.dump(tl, open(os.path.join(P, "%s_resident_timeline.json" % rnd), "w"), indent=1)
    print("profiles/%s_resident_timeline.json" % rnd)
