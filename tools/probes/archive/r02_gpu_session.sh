#!/bin/bash
# GPU session: full GPU test-suite, resident-vs-multi table, profile session, bench lines (developer tool; run through gpurun).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02s4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 300 python tools/bench_resident.py > $O/resident_vs_multi.txt 2>&1
bash tools/r02_profile_session.sh nyu > $O/profile_session.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_driver.log 2>$O/bench_default_driver.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default.log 2>&1
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sparse > $O/bench_sparse.log 2>&1
timeout 300 python bench.py --workload kitti --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_kitti.log 2>&1
timeout 300 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_pac5.log 2>&1
for g in off; do
  timeout 200 python bench.py --workload kitti --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --graph $g > $O/bench_kitti_b1.log 2>&1
  timeout 200 python bench.py --workload nyu --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --graph $g > $O/bench_nyu_b3.log 2>&1
done
CSPN_RESIDENT=off timeout 200 python bench.py --workload kitti --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0 > $O/bench_kitti_b1_multilaunch.log 2>&1
CSPN_RESIDENT=off timeout 200 python bench.py --workload nyu --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0 > $O/bench_nyu_b3_multilaunch.log 2>&1
CSPN_RESIDENT=off timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0 > $O/bench_default_multilaunch.log 2>&1
timeout 600 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train_b3.log 2>&1
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-300; done
