#!/bin/bash
# GPU session 1 of round 2: new tests, per-shard bench lines, training workload (developer tool; run through gpurun).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02s1
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_driver.log 2>$O/bench_default_driver.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default.log 2>&1
for g in off on; do
  timeout 200 python bench.py --workload kitti --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --graph $g > $O/bench_kitti_b1_graph_$g.log 2>&1
  timeout 200 python bench.py --workload nyu --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --graph $g > $O/bench_nyu_b3_graph_$g.log 2>&1
done
timeout 600 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train_b3.log 2>&1
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-1500; done
