#!/bin/bash
# K = 5 resident reverse sweep: parity, then the pac5 training leg before/after.
set -x
mkdir -p gpurun_out/trans
timeout 1500 python -m pytest tests/test_hip_kres.py tests/test_hip_backward.py tests/test_hip_production.py -q -x > gpurun_out/trans/pytest.log 2>&1
tail -15 gpurun_out/trans/pytest.log
CSPN_REVERSE_SWEEP=copy timeout 600 python bench.py --workload pac5 --steps 50 --warmup 10 > gpurun_out/trans/bench_copy.json 2> gpurun_out/trans/bench_copy.err
timeout 600 python bench.py --workload pac5 --steps 50 --warmup 10 > gpurun_out/trans/bench_res.json 2> gpurun_out/trans/bench_res.err
python - <<'PY'
import json
for n in ("copy", "res"):
    try:
        d = json.loads(open("gpurun_out/trans/bench_%s.json" % n).read().strip().splitlines()[-1])
        t = d.get("training_step"); t.pop("roofline", None)
        print(n, d["value"], json.dumps(t))
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trans/prof -o leg -- python $GRAFT_REPO_ROOT/tools/run_train_leg.py --K 5 --dtype f16 --state input --iters 30 > $GRAFT_REPO_ROOT/gpurun_out/trans/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/trans/prof -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200
