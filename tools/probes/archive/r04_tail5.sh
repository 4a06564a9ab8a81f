#!/bin/bash
# Row-split K = 5 backward tail: parity, then A/B of the pac5 training leg on one box.
set -x
mkdir -p gpurun_out/tail5
timeout 1500 python -m pytest tests/test_hip_backward.py tests/test_hip_kres.py tests/test_hip_production.py tests/test_hip_parity.py -q -x > gpurun_out/tail5/pytest.log 2>&1
tail -8 gpurun_out/tail5/pytest.log
for sp in 0 1; do
CSPN_TAIL5_SPLIT=$sp timeout 600 python bench.py --workload pac5 --steps 50 --warmup 10 > gpurun_out/tail5/bench_split$sp.json 2> gpurun_out/tail5/bench_split$sp.err
done
python - <<'PY'
import json
for n in ("split0", "split1"):
    try:
        d = json.loads(open("gpurun_out/tail5/bench_%s.json" % n).read().strip().splitlines()[-1])
        t = d.get("training_step"); t.pop("roofline", None)
        print(n, d["value"], json.dumps(t))
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
for st in input reference; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tail5/prof_$st -o leg -- python $GRAFT_REPO_ROOT/tools/run_train_leg.py --K 5 --dtype f16 --state $st --iters 30 > $GRAFT_REPO_ROOT/gpurun_out/tail5/prof_$st.log 2>&1
head -6 $GRAFT_REPO_ROOT/gpurun_out/tail5/prof_$st/leg_kernel_stats.csv | cut -c1-170
done
