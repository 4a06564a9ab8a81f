#!/bin/bash
# round 4, session 4: fallback / contention tests; kernel stats of the K = 5 fp16-state training leg (what bench.py's pac5 training_step runs)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_resident.py tests/test_hip_kres.py -q -x -m gpu > $O/pytest_res.log 2>&1; echo "pytest rc=$?" >> $O/pytest_res.log
tail -25 $O/pytest_res.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd_pac5_state16 -o bwd -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --state input --iters 30 > $O/stats_bwd_pac5_state16.log 2>&1
f=$(find $O/stats_bwd_pac5_state16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep -i "cspn\|elementwise" $f) > $O/stats_bwd_pac5_state16.csv
python - <<'PY'
import csv,os
p=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out/r04s4/stats_bwd_pac5_state16.csv')
for r in csv.DictReader(open(p)):
    print(r['Calls'], round(float(r['AverageNs'])/1000,1), round(float(r['TotalDurationNs'])/30/1000,1), r['Name'][:120])
PY
