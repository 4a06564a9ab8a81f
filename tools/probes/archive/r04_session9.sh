#!/bin/bash
# round 4, session 9: host fast path of the scored resident forward — tests, host time per call, shard bench lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s9
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_resident.py tests/test_hip_production.py tests/test_hip_parity.py tests/test_abi_and_host.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log | cut -c1-250
python tools/probes/host_overhead_inference.py 2>&1 | grep "host time"
for a in "--workload nyu --batch 3" "--workload kitti --batch 1" "--workload nyu" "--workload kitti"; do
  python bench.py $a --steps 200 --warmup 20 --no-cpu-baseline --no-per-step-leg --no-train-leg --cold-sets 0 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$a', round(d['value']), d['ms_per_step'])"
done
