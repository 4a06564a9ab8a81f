#!/bin/bash
# final evidence pass of round 3: full GPU suite, profile sessions (config 2 / 3), training legs, timelines, sweeps, bench lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${SESSION_TAG:-r03s9}
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log | cut -c1-200
BWD=0 STEP_PLAN=1,48,38,1,512 bash tools/r03_profile_session.sh pac5 > $O/profile_pac5.log 2>&1
bash tools/r03_profile_session.sh nyu > $O/profile_nyu.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd_pac5 -o bwd -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --iters 30 > $O/stats_bwd_pac5.log 2>&1
f=$(find $O/stats_bwd_pac5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep -i "cspn\|elementwise" $f) > $O/stats_bwd_pac5_cspn.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/pac5bwd_fused_$ctr -o pmc -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --iters 6 > $O/pmc_pac5bwd_$ctr.log 2>&1
done
cd $R
export CSPN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null)
python tools/pmc_traffic.py $O/pmc pac5bwd > $O/traffic_pac5bwd.json
python tools/resident_stamps.py 24 228 304 24 > $O/resident_timeline_nyu.txt 2>&1
python tools/resident_stamps.py 3 228 304 24 > $O/resident_timeline_nyu_b3.txt 2>&1
python tools/resident_stamps.py 1 352 1216 24 > $O/resident_timeline_kitti_b1.txt 2>&1
python tools/probes/kres_probe.py stamps plans f32 > $O/kres_probe.txt 2>&1
python tools/probes/resident_s_sweep.py 2>&1 | grep "B=" > $O/resident_s_sweep.txt
python tools/probes/pac3_train_probe.py 2>&1 | grep "B=" > $O/pac3_train_probe.txt
bash tools/r03_bench_lines.sh > $O/bench_lines.log 2>&1; tail -14 $O/bench_lines.log
