#!/bin/bash
# round 4, session 8: same-box A/B of the halo-rows-first thread order of cspnk_d2 (last step of a phase skipped by the halo wavefronts)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kres.py -q -x -m gpu 2>&1 | tail -2 | cut -c1-200
for rep in 1 2; do
  python bench.py --workload pac5 --steps 200 --warmup 20 --no-cpu-baseline --no-per-step-leg --no-train-leg --cold-sets 0 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('row order ON ', round(d['value']), d['ms_per_step'])"
  CSPN_HIP_LIB=$R/_ab/libcspn_norow.so python bench.py --workload pac5 --steps 200 --warmup 20 --no-cpu-baseline --no-per-step-leg --no-train-leg --cold-sets 0 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('row order OFF', round(d['value']), d['ms_per_step'])"
done
python bench.py --workload pac5 --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-per-step-leg --no-train-leg --cold-sets 0 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('B=3 row order ON ', round(d['value']), d['ms_per_step'])"
CSPN_HIP_LIB=$R/_ab/libcspn_norow.so python bench.py --workload pac5 --batch 3 --steps 200 --warmup 20 --no-cpu-baseline --no-per-step-leg --no-train-leg --cold-sets 0 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('B=3 row order OFF', round(d['value']), d['ms_per_step'])"
timeout 300 python tools/probes/kres_probe.py stamps 2>&1 | head -16
