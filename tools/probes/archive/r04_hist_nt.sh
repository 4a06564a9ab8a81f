#!/bin/bash
# non-temporal stores of the history planes in the resident training forms (MODE 2 / 4): same-box A/B, plain and sparse
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/histnt; mkdir -p $O; rm -rf $O/prof*
cd $R; timeout 900 python -m pytest tests/test_hip_backward.py tests/test_training_smoke.py -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for tag in intree histnt; do for sp in "" "--sparse"; do
  lib=$R/cspn_monodepth_amd/ab/libcspn_hip_$tag.so; [ $tag = intree ] && lib=$R/cspn_monodepth_amd/libcspn_hip.so
  CSPN_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}${sp}_$rep -o leg -- python $R/tools/run_train_leg.py --iters 30 $sp > $O/prof_${tag}${sp}_$rep.log 2>&1
done; done; done
python - <<PY
import csv,glob,os
for d in sorted(glob.glob('$O/prof*_*/')):
    f=glob.glob(d+'*kernel_stats.csv')
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        n=r['Name']
        if 'cspn' in n:
            print('%-22s %-52s avg %8.2f us  min %8.2f' % (os.path.basename(d.rstrip('/')), n.replace('(anonymous namespace)::','').replace('void ','')[:52], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
