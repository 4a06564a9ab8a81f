#!/bin/bash
# cspn_grad_tail5 (K = 5 backward tail) with 1 / 2 / 3 / 4 steps of loads in flight (-DCSPN_TAIL5_UNR): rocprofv3 kernel statistics of the fp16
# training leg per variant, two alternating rounds, and a checksum of the gradients (the variants must agree bit for bit)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05tail5; mkdir -p $O; cd $R
for r in 1 2; do for v in "$@"; do
  export CSPN_HIP_LIB=$R/_ab/lib_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_${v}_$r -o bwd -- python tools/run_train_leg.py --K 5 --dtype f16 --state input --iters 40 > $O/s_${v}_$r.log 2>&1
  f=$(find $O/s_${v}_$r -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.reader(open('$f')):
    if 'tail5' in r[0] or 'cspnk_' in r[0]: print('$v round $r: %-50s calls %s avg %.1f us' % (r[0][28:78], r[1], float(r[3])/1000))"
  rm -rf $O/s_${v}_$r
done; done
