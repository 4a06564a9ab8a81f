#!/bin/bash
# Round-3 GPU session 2: K x K resident kernel bring-up (tests), DDP diagnosis, training-leg A/B of the backward check.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s2
mkdir -p $O
cd $R
timeout 1200 python -m pytest -q -x tests/test_hip_kres.py > $O/pytest_kres.log 2>&1; echo "rc=$?" >> $O/pytest_kres.log
tail -25 $O/pytest_kres.log
timeout 600 python -m pytest -q tests/test_hip_resident.py tests/test_hip_production.py tests/test_hip_parity.py -x > $O/pytest_res.log 2>&1; echo "rc=$?" >> $O/pytest_res.log
tail -4 $O/pytest_res.log
timeout 300 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_pac5.log 2>$O/bench_pac5.err
tail -1 $O/bench_pac5.log | cut -c1-300; tail -3 $O/bench_pac5.err
CSPN_RESIDENT=off timeout 300 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0 > $O/bench_pac5_multi.log 2>&1
tail -1 $O/bench_pac5_multi.log | cut -c1-200
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | cut -c1-200
python - <<'PY' 2>&1 | tail -5
import json
for f in ("bench_default","bench_pac5"):
    try:
        d=json.loads(open("/root/repo/gpurun_out/r03s2/%s.log"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("training_step",{}).get("fwd_bwd_us"), d["default_schedule"]["kernel"][:60])
    except Exception as e: print(f, "ERR", e)
PY
# DDP worker directly, with its stderr
env PYTHONPATH=$R OMP_NUM_THREADS=4 CSPN_RESIDENT=off timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 tests/dist_ddp_worker.py gloo 2 > $O/ddp_worker.log 2>&1; echo "rc=$?" >> $O/ddp_worker.log
grep -v "^\[Gloo\]\|^W0\|^\*\*\*" $O/ddp_worker.log | tail -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pac5 -o bench -- python $R/bench.py --workload pac5 --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0 --prewarm-s 0 > $O/stats_pac5.log 2>&1
f=$(find $O/stats_pac5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep cspn $f) > $O/stats_pac5_cspn.csv; cut -c1-160 $O/stats_pac5_cspn.csv
