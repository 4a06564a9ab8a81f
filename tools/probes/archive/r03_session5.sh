#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s5
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log | cut -c1-250
BWD=0 STEP_PLAN=1,48,38,1,512 bash tools/r03_profile_session.sh pac5 > $O/profile_pac5.log 2>&1; tail -5 $O/profile_pac5.log
timeout 300 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline > $O/r03_bench_pac5.json 2>$O/bench_pac5.err
tail -1 $O/r03_bench_pac5.json | cut -c1-250
