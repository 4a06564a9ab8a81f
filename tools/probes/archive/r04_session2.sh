#!/bin/bash
# round 4, session 2: the dot-product form of the K x K resident launch — parity tests, same-box A/B against the FMA form, timeline
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s2
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_kres.py -q -x -m gpu > $O/pytest_kres.log 2>&1; echo "pytest rc=$?" >> $O/pytest_kres.log
tail -15 $O/pytest_kres.log | cut -c1-300
run() { n=$1; shift; timeout 400 "$@" 2>$O/$n.err | tail -1 > $O/bench_$n.json; }
run pac5_dot2 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg
CSPN_KRES_STEP=fma run pac5_fma python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg
run pac5_dot2_b3 python bench.py --workload pac5 --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg
CSPN_KRES_STEP=fma run pac5_fma_b3 python bench.py --workload pac5 --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg
for f in $O/bench_*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']), d['ms_per_step'])")"; done
timeout 300 python tools/probes/kres_probe.py stamps > $O/kres_stamps.txt 2>&1; head -40 $O/kres_stamps.txt
