#!/bin/bash
# bench.py's training leg is bimodal per PROCESS on one box (205-222 or 265-285 us per pass): which kernels are slower in the slow mode?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bimodal; mkdir -p $O; rm -rf $O/p*
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$i -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --cold-sets 0 --no-per-step-leg > $O/p$i.json 2> $O/p$i.err
  python - <<PY
import json,csv,glob
d=json.loads(open('$O/p$i.json').read().strip().splitlines()[-1])
ks={}
for r in csv.DictReader(open(glob.glob('$O/p$i/*kernel_stats.csv')[0])):
    n=r['Name']
    if 'grad_tail' in n or ', 2, 1, 0>' in n or ', 4, 1, 0>' in n or ', 1, 1, 0>' in n: ks[n.replace('(anonymous namespace)::','').replace('void ','')[:42]]=round(float(r['AverageNs'])/1e3,1)
print('run $i: value', round(d['value']), 'train', round(d['training_step']['fwd_bwd_us'],1), ks)
PY
done
