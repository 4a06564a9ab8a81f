#!/bin/bash
# Volume-free 3x3 training path (forward publishes S only; reverse sweep and tail rebuild the taps): parity, then same-box A/B
# against CSPN_TRAIN_VOLUME=1 (rocprofv3 kernel times of the training leg, bench training_step)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/volfree
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_resident.py tests/test_hip_backward.py tests/test_training_smoke.py tests/test_abi_and_host.py -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do for sp in "" "--sparse"; do
  CSPN_TRAIN_VOLUME=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vol$v$sp -o leg -- python $R/tools/run_train_leg.py --iters 30 $sp > $O/prof_vol$v$sp.log 2>&1
done; done
cd $R
for v in 0 1; do
CSPN_TRAIN_VOLUME=$v timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench_vol$v.json 2> $O/bench_vol$v.err
done
python - <<PY
import csv,glob,os,json
for d in sorted(glob.glob('$O/prof_*/')):
    f=glob.glob(d+'*kernel_stats.csv')
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        n=r['Name']
        if 'cspn' in n:
            print('%-22s %-60s avg %8.2f us  min %8.2f' % (os.path.basename(d.rstrip('/')), n.replace('(anonymous namespace)::','').replace('void ','')[:60], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
for v in (0,1):
    try:
        d=json.loads(open('$O/bench_vol%d.json'%v).read().strip().splitlines()[-1])
        t=d['training_step']; t.pop('roofline',None)
        print('vol',v,d['value'],json.dumps(t))
    except Exception as e: print('bench',v,'failed',e)
PY
