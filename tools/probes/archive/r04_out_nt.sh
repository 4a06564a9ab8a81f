#!/bin/bash
# resident inference launch: non-temporal store of the refined depth (1) / load of the metrics target (2): bench value, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2 3; do for tag in intree outnt1 outnt3; do
  lib=$R/cspn_monodepth_amd/ab/libcspn_hip_$tag.so; [ $tag = intree ] && lib=$R/cspn_monodepth_amd/libcspn_hip.so
  for wl in nyu kitti; do
  CSPN_HIP_LIB=$lib python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-train-leg --cold-sets 0 --no-per-step-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag $wl rep$rep', round(d['value']), round(d['ms_per_step']*1e3,2))"
  done
done; done
