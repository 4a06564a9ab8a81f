#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes written by tools/r02_profile_session.sh into per-kernel averages and the
HBM-traffic figure bench.py reports as roofline.traffic (profiles/traffic_<workload>.json).

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so wide
coalesced reads are under-counted by 2x (MI355X_MICROARCH.md §HBM) -> hbm_bytes = (2*FETCH + WRITE)*1024.
The S=1 ("step") passes calibrate that correction against a known byte count."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root, wl = sys.argv[1], sys.argv[2]
res = defaultdict(lambda: defaultdict(dict))   # sched -> kernel -> counter -> avg
for d in sorted(glob.glob(os.path.join(root, wl + "_*"))):
    if not os.path.isdir(d):
        continue
    sched = os.path.basename(d)[len(wl) + 1:].split("_")[0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*$", "", k).replace("void ", "")
            if not k.startswith("cspn"):
                continue
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        for (k, c), (tot, n) in acc.items():
            res[sched][k][c] = tot / n
            res[sched][k]["_dispatches_" + c] = n
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cspn_monodepth_amd import _lib as _cl   # noqa: E402  (source digest only; nothing is launched)
out = {"workload": wl, "per_kernel": res, "source_digest": _cl.code_digest(),
       "commit": os.environ.get("CSPN_COMMIT", "")}
for sched in res:
    for k, v in res[sched].items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            v["hbm_bytes_corrected"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
dom = [(k, v) for k, v in res.get("fused", {}).items()
       if k.startswith(("cspn_prop_fused", "cspn3_resident", "cspnk_resident", "cspnk_d2")) and "hbm_bytes_corrected" in v]
if dom:
    # the default schedule launches two instances of the propagation kernel per forward (the first derives and
    # publishes the weights, the others stream them): bench.py averages over all launches, so does this figure
    n = sum(v["_dispatches_FETCH_SIZE"] for _, v in dom)
    out["dominant_kernel"] = max(dom, key=lambda kv: kv[1]["_dispatches_FETCH_SIZE"])[0]
    out["launch_mix"] = {k: v["_dispatches_FETCH_SIZE"] / n for k, v in dom}
    out["hbm_bytes_per_launch"] = sum(v["hbm_bytes_corrected"] * v["_dispatches_FETCH_SIZE"] for _, v in dom) / n
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(os.path.dirname(root), "pmc_out"), exist_ok=True)
json.dump(out, open(os.path.join(os.path.dirname(root), "pmc_out", "traffic_%s.json" % wl), "w"), indent=1)
