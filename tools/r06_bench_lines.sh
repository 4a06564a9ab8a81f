#!/bin/bash
# The committed bench lines of round 6 (run through gpurun; tools/collect_profiles.py r06 <session> copies gpurun_out/r06lines/*.json into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06lines
mkdir -p $O
cd $R
run() { n=$1; shift; timeout 400 "$@" 2>$O/$n.err | tail -1 > $O/r06_bench_$n.json; }
run default_driver python bench.py --steps 20 --warmup 5
run default python bench.py --steps 200 --warmup 20 --no-cpu-baseline
CSPN_RESIDENT=off run default_multilaunch python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --cold-sets 0 --no-sparse-leg
run sparse python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sparse
run kitti python bench.py --workload kitti --steps 100 --warmup 10 --no-cpu-baseline
run pac5 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline
CSPN_RESIDENT=off run pac5_multilaunch python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0 --no-sparse-leg
run kitti_b1 python bench.py --workload kitti --batch 1 --steps 200 --warmup 20 --no-cpu-baseline
run nyu_b3 python bench.py --workload nyu --batch 3 --steps 200 --warmup 20 --no-cpu-baseline
run nyu_b8 python bench.py --workload nyu --batch 8 --steps 200 --warmup 20 --no-cpu-baseline
run pac5_b3 python bench.py --workload pac5 --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg
run train_b3 python bench.py --workload train --steps 30 --warmup 5 --infer-batch 24
run train_b3_graph python bench.py --workload train --steps 30 --warmup 5 --graph on
for f in $O/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']), d['ms_per_step'], (d.get('training_step') or {}).get('fwd_bwd_us'), (d.get('sparse_variant') or {}).get('value'))")"; done
