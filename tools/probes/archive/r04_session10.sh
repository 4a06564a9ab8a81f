#!/bin/bash
# round 4, session 10: "data is the flag" exchange — correctness (resident / kres / backward / c-host tests, stress), same-box A/B against the previous commit
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s10
mkdir -p $O
cd $R
timeout 2000 python -m pytest tests/test_hip_resident.py tests/test_hip_kres.py tests/test_c_host.py tests/test_hip_backward.py tests/test_hip_production.py -q -x -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log | cut -c1-250
timeout 600 python tools/probes/resident_stress.py 4000 > $O/stress.txt 2>&1; tail -5 $O/stress.txt
bench() { python bench.py "$@" --no-cpu-baseline --no-per-step-leg --cold-sets 0 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), round(d['ms_per_step']*1e3,2), (d.get('training_step') or {}).get('fwd_bwd_us'))"; }
for rep in 1 2; do
for a in "--workload nyu --steps 200 --warmup 20" "--workload pac5 --steps 200 --warmup 20" "--workload nyu --batch 3 --steps 200 --warmup 20 --no-train-leg" "--workload kitti --batch 1 --steps 200 --warmup 20 --no-train-leg" "--workload kitti --steps 100 --warmup 10 --no-train-leg" "--workload pac5 --batch 3 --steps 200 --warmup 20 --no-train-leg"; do
  echo "NEW  $a: $(bench $a)"
  echo "PREV $a: $(cd $R/_ab/prev && bench $a)"
done
done
