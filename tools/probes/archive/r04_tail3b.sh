#!/bin/bash
# 3x3 backward tail: block size / unroll variants and the batch-size dependence (1622 workgroups at B = 24 are 1.58 rounds of the
# 1024 that fit the chip at 4 wavefronts per SIMD)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tail3b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for tag in intree tail_b128 tail_b64 tail_unr2 tail_unr4; do
  lib=$R/cspn_monodepth_amd/ab/libcspn_hip_$tag.so; [ $tag = intree ] && lib=$R/cspn_monodepth_amd/libcspn_hip.so
  for bb in 24 15 30; do
  [ $tag != intree ] && [ $bb != 24 ] && continue
  CSPN_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_b$bb -o leg -- python $R/tools/run_train_leg.py --iters 30 --batch $bb > $O/prof_${tag}_b$bb.log 2>&1
  done
done
python - <<PY
import csv,glob,os
for d in sorted(glob.glob('$O/prof_*/')):
    f=glob.glob(d+'*kernel_stats.csv')
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        n=r['Name']
        if 'cspn' in n:
            print('%-22s %-60s avg %8.2f us  min %8.2f' % (os.path.basename(d.rstrip('/')), n.replace('(anonymous namespace)::','').replace('void ','')[:60], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
