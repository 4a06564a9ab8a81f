#!/bin/bash
# round 4, session 5: contention probe; backward parity after the raw-load rework of the fp16 tail; kernel stats of the K = 5 fp16-state training leg
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s5
mkdir -p $O
cd $R
python tools/probes/contention_probe.py 2>&1 | tail -16
timeout 1500 python -m pytest tests/test_hip_backward.py tests/test_hip_production.py tests/test_hip_parity.py -q -x -m gpu > $O/pytest_bwd.log 2>&1; echo "pytest rc=$?" >> $O/pytest_bwd.log
tail -4 $O/pytest_bwd.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd_pac5_state16 -o bwd -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --state input --iters 30 > $O/stats_bwd_pac5_state16.log 2>&1
f=$(find $O/stats_bwd_pac5_state16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep -i "cspn\|elementwise" $f) > $O/stats_bwd_pac5_state16.csv
python - <<'PY'
import csv,os
p=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out/r04s5/stats_bwd_pac5_state16.csv')
for r in csv.DictReader(open(p)):
    print(r['Calls'], round(float(r['AverageNs'])/1000,1), round(float(r['TotalDurationNs'])/30/1000,1), r['Name'][:120])
PY
cd $R
python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-per-step-leg --cold-sets 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('pac5', round(d['value']), d['ms_per_step'], d['training_step']['fwd_bwd_us'])"
