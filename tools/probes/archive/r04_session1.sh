#!/bin/bash
# round 4, session 1: VALU issue rates of the candidate instructions for the K x K step + same-box baselines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s1
mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/valu_rate_probe tools/probes/valu_rate_probe.hip && timeout 300 /tmp/valu_rate_probe > $O/valu_rate_probe.txt 2>&1
cat $O/valu_rate_probe.txt
run() { n=$1; shift; timeout 400 "$@" 2>$O/$n.err | tail -1 > $O/bench_$n.json; }
run pac5 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline
run default python bench.py --steps 200 --warmup 20 --no-cpu-baseline
run kitti_b1 python bench.py --workload kitti --batch 1 --steps 200 --warmup 20 --no-cpu-baseline
run nyu_b3 python bench.py --workload nyu --batch 3 --steps 200 --warmup 20 --no-cpu-baseline
for f in $O/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']), d['ms_per_step'], (d.get('training_step') or {}).get('fwd_bwd_us'))")"; done
