#!/usr/bin/env python3
"""Summarise the SQ counter passes of tools/r02_profile_session.sh into profiles/<tag>_sq_<workload>.json:
per kernel instance the average counter values and the derived busy fractions."""
import csv, glob, json, os, re, sys, collections
root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"\(.*$", "", k).replace("void ", "")
        if not k.startswith("cspn"):
            continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
res = {}
for k, v in acc.items():
    c = {n: t / cnt for n, (t, cnt) in v.items()}
    c["_dispatches"] = max(cnt for _, cnt in v.values())
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for name, key in (("valu_frac_of_wave_cycles", "SQ_ACTIVE_INST_VALU"), ("lds_frac_of_wave_cycles", "SQ_ACTIVE_INST_LDS"),
                          ("wait_any_frac_of_wave_cycles", "SQ_WAIT_ANY"), ("wait_inst_lds_frac_of_wave_cycles", "SQ_WAIT_INST_LDS")):
            if key in c:
                c[name] = c[key] / wc
    if c.get("SQ_LDS_IDX_ACTIVE"):
        c["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    res[k] = c
json.dump({"per_kernel": res, "note": "rocprofv3 --pmc SQ_* passes (one counter group per run), averages per dispatch; "
           "fractions are relative to SQ_WAVE_CYCLES (wave-resident cycles summed over waves)"}, open(out, "w"), indent=1)
print(json.dumps({k: {n: round(x, 4) for n, x in v.items() if "frac" in n} for k, v in res.items()}, indent=1))
