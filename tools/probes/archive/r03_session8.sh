#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s8
mkdir -p $O
cd $R
timeout 900 python -m pytest -q -x tests/test_hip_resident.py tests/test_hip_kres.py tests/test_hip_backward.py > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-per-step-leg --cold-sets 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-like', round(d['value']), d['ms_per_step'], d['training_step']['fwd_bwd_us'])"; done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-per-step-leg --cold-sets 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('200 steps', round(d['value']), d['ms_per_step'], d['training_step']['fwd_bwd_us'])"
timeout 300 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-per-step-leg --cold-sets 0 --no-train-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pac5', round(d['value']), d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-per-step-leg --cold-sets 0 --no-train-leg --prewarm-s 0 > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); grep resident $f | cut -c1-150
