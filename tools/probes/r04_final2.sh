#!/bin/bash
# round 4: PMC / kernel-stat passes for the KITTI and sparse workloads, then their bench lines again (so that they quote fresh traffic)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
BWD=0 STEP_PLAN=1,64,44,1,1024 bash tools/r04_profile_session.sh kitti > gpurun_out/profile_kitti.log 2>&1
BWD=0 EXTRA=--sparse bash tools/r04_profile_session.sh nyu nyu_sparse > gpurun_out/profile_nyu_sparse.log 2>&1
cp gpurun_out/r04prof_kitti/traffic_kitti.json profiles/r04_pmc_traffic_kitti.json
cp gpurun_out/r04prof_nyu_sparse/traffic_nyu_sparse.json profiles/r04_pmc_traffic_nyu_sparse.json
O=$R/gpurun_out/r04lines2; mkdir -p $O
run() { n=$1; shift; timeout 400 "$@" 2>$O/$n.err | tail -1 > $O/r04_bench_$n.json; }
run sparse python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sparse
run kitti python bench.py --workload kitti --steps 100 --warmup 10 --no-cpu-baseline
run kitti_b1 python bench.py --workload kitti --batch 1 --steps 200 --warmup 20 --no-cpu-baseline
cp profiles/r04_pmc_traffic_kitti.json profiles/r04_pmc_traffic_nyu_sparse.json $O/
for f in $O/r04_bench_*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']), d['ms_per_step'], d['default_schedule'].get('traffic_stale'))")"; done
