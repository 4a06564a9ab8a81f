#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s3
mkdir -p $O
cd $R
timeout 600 python tools/probes/kres_probe.py diff stamps sweep > $O/kres_probe.log 2>&1; tail -60 $O/kres_probe.log
env PYTHONPATH=$R OMP_NUM_THREADS=4 CSPN_RESIDENT=off timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 tests/dist_ddp_worker.py gloo 2 > $O/ddp_worker.log 2>&1; echo "rc=$?" >> $O/ddp_worker.log
grep "DDP_CHECK\|AssertionError\|rc=" $O/ddp_worker.log | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd -o bwd -- python $R/tools/run_train_leg.py --iters 30 > $O/stats_bwd.log 2>&1
f=$(find $O/stats_bwd -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep cspn $f) > $O/stats_bwd_cspn.csv; cut -c1-150 $O/stats_bwd_cspn.csv
cd $R
for chk in on off; do CSPN_BWD_CHECK=$chk timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-per-step-leg --cold-sets 0 > $O/bench_chk_$chk.log 2>&1; python -c "
import json;d=json.loads(open('$O/bench_chk_$chk.log').read().strip().splitlines()[-1]);print('$chk',d['value'],d['training_step']['fwd_bwd_us'])"; done
