#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s6
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_resident.py tests/test_hip_kres.py tests/test_distributed_gpu.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log | cut -c1-250
timeout 900 python tools/probes/resident_stress.py 8000 sparse > $O/stress_sparse.txt 2>&1; tail -6 $O/stress_sparse.txt
timeout 600 python tools/probes/resident_stress.py 8000 > $O/stress.txt 2>&1; tail -5 $O/stress.txt
