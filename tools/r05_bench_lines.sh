#!/bin/bash
# The committed bench lines of round 5 (run through gpurun; copy gpurun_out/r05lines/*.json into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05lines
mkdir -p $O
cd $R
run() { n=$1; shift; timeout 400 "$@" 2>$O/$n.err | tail -1 > $O/r05_bench_$n.json; }
run default_driver python bench.py --steps 20 --warmup 5
run default python bench.py --steps 200 --warmup 20 --no-cpu-baseline
run default_bind_off python bench.py --steps 200 --warmup 20 --no-cpu-baseline --cpu-bind off
run default_driver_bind_off python bench.py --steps 20 --warmup 5 --cpu-bind off
run default_bind_auto python bench.py --steps 200 --warmup 20 --no-cpu-baseline --cpu-bind auto
CSPN_RESIDENT=off run default_multilaunch python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --cold-sets 0
run sparse python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sparse
run kitti python bench.py --workload kitti --steps 100 --warmup 10 --no-cpu-baseline
run pac5 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline
CSPN_RESIDENT=off run pac5_multilaunch python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0
run kitti_b1 python bench.py --workload kitti --batch 1 --steps 200 --warmup 20 --no-cpu-baseline
run nyu_b3 python bench.py --workload nyu --batch 3 --steps 200 --warmup 20 --no-cpu-baseline
run pac5_b3 python bench.py --workload pac5 --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg
CSPN_RESIDENT=off run pac5_b3_multilaunch python bench.py --workload pac5 --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0
run train_b3 python bench.py --workload train --steps 30 --warmup 5
run train_b3_graph python bench.py --workload train --steps 30 --warmup 5 --graph on
run train_b3_round2_settings python bench.py --workload train --steps 30 --warmup 5 --conv-db off --sgd foreach
for f in $O/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']), d['ms_per_step'], (d.get('training_step') or {}).get('fwd_bwd_us'))")"; done
