#!/usr/bin/env python3
"""Copy the judged summaries of the last GPU session from gpurun_out/ (scratch) into profiles/ (tracked)."""
import csv, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
for src, dst in (("bench_default.log", "bench_nyu_default.json"), ("bench_sparse.log", "bench_nyu_sparse.json"),
                 ("bench_kitti.log", "bench_kitti.json"), ("bench_pac5.log", "bench_pac5_fp16.json"),
                 ("bench_auto.log", "bench_nyu_autotuned.json")):
    f = os.path.join(G, src)
    if os.path.exists(f):
        line = open(f).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(P, "%s_%s" % (tag, dst)), "w").write(line + "\n")
ks = os.path.join(G, "prof_bench", "bench_kernel_stats.csv")
if os.path.exists(ks):
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(P, "%s_bench_nyu_kernel_stats.csv" % tag), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline  (MI355X)\n")
        cols = ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")
        f.write(",".join(cols) + "\n")
        for r in rows:
            if "cspn" in r["Name"]:
                f.write(",".join('"%s"' % r[k] if k == "Name" else r[k] for k in cols) + "\n")
for w in ("nyu", "kitti", "pac5"):
    t = os.path.join(G, "pmc_out", "traffic_%s.json" % w)
    if os.path.exists(t):
        shutil.copy(t, os.path.join(P, "traffic_%s.json" % w))
        shutil.copy(t, os.path.join(P, "%s_pmc_traffic_%s.json" % (tag, w)))
for w in ("nyu", "kitti", "pac5", "nyu_sparse"):
    f = os.path.join(G, "tune_%s.log" % w)
    if os.path.exists(f):
        keep = [l for l in open(f) if any(k in l for k in ("prepare", "best:", "module forward", "swept"))]
        open(os.path.join(P, "%s_plan_sweep_%s.txt" % (tag, w)), "w").writelines(keep)
print(sorted(os.listdir(P)))
