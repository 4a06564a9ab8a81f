#!/bin/bash
# round 4, session 3: dot-product form after the hazard fix — A/B bench, timeline, full GPU suite
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s3
mkdir -p $O
cd $R
run() { n=$1; shift; timeout 400 "$@" 2>$O/$n.err | tail -1 > $O/bench_$n.json; }
run pac5_dot2 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg
CSPN_KRES_STEP=fma run pac5_fma python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg
run pac5_dot2_b3 python bench.py --workload pac5 --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg
for f in $O/bench_*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']), d['ms_per_step'], d.get('metrics_check'))")"; done
timeout 300 python tools/probes/kres_probe.py stamps > $O/kres_stamps.txt 2>&1; head -16 $O/kres_stamps.txt
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log | cut -c1-300
