#!/bin/bash
# SQ-level counters of the bench kernels (one pass, 8 SQ counters), developer tool.
set -u
EXTRA="${1:-}"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -o pmc -- \
    python $R/bench.py $EXTRA --steps 6 --warmup 2 --no-cpu-baseline --prewarm-s 0 --cold-sets 0 --no-per-step-leg --no-train-leg --graph off > $O/$tag.log 2>&1
done
cd $R
python - <<PY
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"\(.*$", "", k).replace("void ", "")
        if not k.startswith("cspn"): continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in acc.items():
    print(k)
    print("   " + "  ".join("%s=%.0f" % (c.replace("SQ_", ""), t / n) for c, (t, n) in sorted(v.items())))
PY
