#!/bin/bash
# 3x3 backward tail: parity of the rebuilt-weights epilogue, then per-kernel times (rocprofv3) of four builds of the tail:
# in-tree (w rebuilt from the guidance), base (w read from the tap volume), probe1 (no scatter epilogue), probe2 (no stream loop);
# and kernel-level times of the stride-2 PAC rows (bench_pac_conv's own numbers are host-bound below ~20 us per call).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tail3
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_backward.py tests/test_training_smoke.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for tag in intree tail_base tail_probe1 tail_probe2; do
  lib=$R/cspn_monodepth_amd/ab/libcspn_hip_$tag.so; [ $tag = intree ] && lib=$R/cspn_monodepth_amd/libcspn_hip.so
  for sp in "" "--sparse"; do
  CSPN_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag$sp -o leg -- python $R/tools/run_train_leg.py --iters 30 $sp > $O/prof_$tag$sp.log 2>&1
  echo "== $tag $sp"; grep -i "cspn" $O/prof_$tag$sp/leg_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
  done
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s2 -o s2 -- python $R/tools/bench_pac_conv.py --only stride2 > $O/prof_s2.log 2>&1
grep -i "pac_s2" $O/prof_s2/s2_kernel_stats.csv | cut -d, -f1-4 | cut -c1-170
