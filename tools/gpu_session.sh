#!/bin/bash
# One GPU-box session: tests, plan sweeps, bench, rocprofv3 kernel stats (developer tool; run through gpurun).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for wl in nyu kitti pac5; do
  timeout 300 python tools/tune.py --workload $wl --out $O/tune_$wl.jsonl > $O/tune_$wl.log 2>&1
done
timeout 300 python tools/tune.py --workload nyu --sparse --out $O/tune_nyu_sparse.jsonl > $O/tune_nyu_sparse.log 2>&1
timeout 300 python bench.py --steps 200 --warmup 20 > $O/bench_default.log 2>&1
timeout 300 python bench.py --steps 200 --warmup 20 --plan auto --no-cpu-baseline > $O/bench_auto.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg > $O/prof_bench.log 2>&1
cd $R
tail -3 $O/pytest_gpu.log
for wl in nyu kitti pac5 nyu_sparse; do echo "== $wl"; grep -E "prepare|best:|module forward" $O/tune_$wl.log; done
tail -1 $O/bench_default.log; tail -1 $O/bench_auto.log
ls $O/prof_bench
