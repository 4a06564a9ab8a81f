#!/bin/bash
# backward tails: non-temporal loads of the d history (1) / G history (2), non-temporal epilogue stores (4) / epilogue guidance loads (8)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/tailnt; mkdir -p $O; rm -rf $O/prof_*
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for tag in ${TAGS:-intree nt2 nt10 nt14}; do
  lib=$R/cspn_monodepth_amd/ab/libcspn_hip_$tag.so; [ $tag = intree ] && lib=$R/cspn_monodepth_amd/libcspn_hip.so
  CSPN_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_$rep -o leg -- python $R/tools/run_train_leg.py --iters 30 > $O/prof_${tag}_$rep.log 2>&1
  CSPN_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5_${tag}_$rep -o leg -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --state input --iters 30 > $O/prof5_${tag}_$rep.log 2>&1
done; done
python - <<PY
import csv,glob,os
for d in sorted(glob.glob('$O/prof*_*/')):
    f=glob.glob(d+'*kernel_stats.csv')
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        n=r['Name']
        if 'grad_tail' in n:
            print('%-16s %-52s avg %8.2f us  min %8.2f' % (os.path.basename(d.rstrip('/')), n.replace('(anonymous namespace)::','').replace('void ','')[:52], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
