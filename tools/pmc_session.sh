#!/bin/bash
# PMC passes (separate runs per counter group, kernel-trace only — see MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc
WL=${1:-nyu}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $ctr | tr ' ' '+')
  STEP_PLAN=${2:-"1,32,29,1,256"}
  for plan in default "$STEP_PLAN"; do
    ptag=$( [ "$plan" = default ] && echo fused || echo step )
    extra=""; [ "$plan" = default ] || extra="--plan $plan"
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${WL}_${ptag}_$tag -o pmc -- \
      python $R/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0 --prewarm-s 0 --graph off $extra > $O/${WL}_${ptag}_$tag.log 2>&1
  done
done
cd $R
python tools/pmc_traffic.py $O $WL
