#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s7
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log | cut -c1-220
