#!/bin/bash
# round 4, session 7: K = 5 fp16 training forward on the dot-product kernel; contention test with four tenant streams; scale-sweep dry run
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s7
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_kres.py tests/test_hip_resident.py tests/test_hip_backward.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log | cut -c1-250
timeout 900 python -m pytest tests/test_distributed_gpu.py -q -x -m gpu -k "scale_sweep or kitti" > $O/pytest_dist.log 2>&1; tail -3 $O/pytest_dist.log | cut -c1-250
python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-per-step-leg --cold-sets 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('pac5', round(d['value']), d['ms_per_step'], d['training_step']['fwd_bwd_us'])"
CSPN_RESIDENT=off python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-per-step-leg --cold-sets 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('pac5 resident off', round(d['value']), d['ms_per_step'], d['training_step']['fwd_bwd_us'])"
