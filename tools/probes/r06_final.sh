#!/bin/bash
# final evidence pass of round 6: full GPU suite, smoke,
# profile sessions of all four bench workloads + both training legs, PMC summaries copied into profiles/ ON THE BOX so that the
# bench lines that follow quote fresh traffic, timelines, bench lines, PAC op table, scale-sweep dry run
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${SESSION_TAG:-r06final}
mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log | cut -c1-200
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/r06_profile_session.sh nyu > $O/profile_nyu.log 2>&1
BWD=0 STEP_PLAN=1,48,38,1,512 bash tools/r06_profile_session.sh pac5 > $O/profile_pac5.log 2>&1
BWD=0 STEP_PLAN=1,64,44,1,1024 bash tools/r06_profile_session.sh kitti > $O/profile_kitti.log 2>&1
BWD=0 EXTRA=--sparse bash tools/r06_profile_session.sh nyu nyu_sparse > $O/profile_nyu_sparse.log 2>&1
cd /tmp && export TMPDIR=/tmp
for st in input reference; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd_pac5_$st -o bwd -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --state $st --iters 30 > $O/stats_bwd_pac5_$st.log 2>&1
  f=$(find $O/stats_bwd_pac5_$st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep -i "cspn\|elementwise" $f) > $O/stats_bwd_pac5_${st}_cspn.csv
done
cp $O/stats_bwd_pac5_input_cspn.csv $O/stats_bwd_pac5_cspn.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/pac5bwd_fused_$ctr -o pmc -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --state input --iters 6 > $O/pmc_pac5bwd_$ctr.log 2>&1
done
cd $R
export CSPN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null)
python tools/pmc_traffic.py $O/pmc pac5bwd > $O/traffic_pac5bwd.json
for t in nyu pac5 kitti nyu_sparse; do cp gpurun_out/r06prof_$t/traffic_$t.json profiles/r06_pmc_traffic_$t.json; cp gpurun_out/r06prof_$t/sq_$t.json profiles/r06_sq_$t.json 2>/dev/null; done
cp gpurun_out/r06prof_nyu/traffic_nyubwd.json profiles/r06_pmc_traffic_nyubwd.json
cp $O/traffic_pac5bwd.json profiles/r06_pmc_traffic_pac5bwd.json
python tools/resident_stamps.py 24 228 304 24 > $O/resident_timeline_nyu.txt 2>&1
python tools/resident_stamps.py 3 228 304 24 > $O/resident_timeline_nyu_b3.txt 2>&1
python tools/resident_stamps.py 1 352 1216 24 > $O/resident_timeline_kitti_b1.txt 2>&1
python tools/probes/kres_probe.py stamps > $O/kres_probe.txt 2>&1
bash tools/r06_bench_lines.sh > $O/bench_lines.log 2>&1; tail -16 $O/bench_lines.log
timeout 900 python tools/bench_pac_conv.py --json $O/pac_conv_unpool.json > $O/pac_conv_unpool.log 2>&1; cat $O/pac_conv_unpool.log | cut -c1-170
GPUS="1 2" BACKEND=gloo STEPS=20 WORKLOADS="nyu kitti pac5" OUT=$O/scale bash tools/scale_sweep.sh > $O/scale_sweep.txt 2>&1; tail -8 $O/scale_sweep.txt
