#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s4
mkdir -p $O
cd $R
timeout 1200 python -m pytest -q -x tests/test_hip_kres.py > $O/pytest_kres.log 2>&1; echo "rc=$?" >> $O/pytest_kres.log
tail -12 $O/pytest_kres.log
timeout 600 python tools/probes/kres_probe.py weights stamps > $O/kres_probe.log 2>&1; grep -v "^pattern" $O/kres_probe.log | tail -45
timeout 900 python -m pytest -q -x tests/test_hip_parity.py tests/test_hip_production.py tests/test_hip_backward.py tests/test_hip_fuzz.py > $O/pytest_pac.log 2>&1; tail -3 $O/pytest_pac.log
env PYTHONPATH=$R OMP_NUM_THREADS=4 CSPN_RESIDENT=off timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 tests/dist_ddp_worker.py gloo 2 > $O/ddp_worker.log 2>&1; echo "rc=$?" >> $O/ddp_worker.log
grep "DDP_CHECK\|AssertionError\|rc=" $O/ddp_worker.log | tail -5
