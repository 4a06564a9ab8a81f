#!/bin/bash
# Round-6 profiling session (same passes as rounds 2 and 3) (run through gpurun): rocprofv3 kernel stats + PMC traffic + SQ counters of the bench command,
# isolated S=1 kernel stats, backward kernels.  Counter passes are separate runs with --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06prof_${2:-${1:-nyu}}
WL=${1:-nyu}
TAG=${2:-$WL}            # name of the summary (e.g. nyu_sparse with EXTRA=--sparse)
EXTRA=${EXTRA:-}
BWD=${BWD:-1}            # 0: skip the passes over tools/run_train_leg.py (they are NYU config 2 only)
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload $WL $EXTRA --no-cpu-baseline --no-train-leg --no-sparse-leg --no-stock-ops-leg --cold-sets 0 --prewarm-s 0 --graph off"
# 1. kernel stats of the bench command (default schedule + S=1 leg), and of the S=1 schedule alone
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o bench -- $B --steps 50 --warmup 10 > $O/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_s1 -o bench -- $B --steps 50 --warmup 10 --no-per-step-leg --plan ${STEP_PLAN:-1,64,57,1,1024} > $O/stats_s1.log 2>&1
CSPN_RESIDENT=off rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_multi -o bench -- $B --steps 50 --warmup 10 --no-per-step-leg > $O/stats_multi.log 2>&1
[ "$BWD" = 1 ] && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd -o bwd -- python $R/tools/run_train_leg.py --iters 30 > $O/stats_bwd.log 2>&1
# 2. HBM traffic (FETCH_SIZE / WRITE_SIZE + L2 hit/miss), one counter group per run
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $ctr | tr ' ' '+')
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/${TAG}_fused_$tag -o pmc -- $B --steps 6 --warmup 2 --no-per-step-leg > $O/pmc_fused_$tag.log 2>&1
  CSPN_RESIDENT=off rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/${TAG}_multi_$tag -o pmc -- $B --steps 6 --warmup 2 --no-per-step-leg > $O/pmc_multi_$tag.log 2>&1
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/${TAG}_step_$tag -o pmc -- $B --steps 6 --warmup 2 --no-per-step-leg --plan ${STEP_PLAN:-1,64,57,1,1024} > $O/pmc_step_$tag.log 2>&1
  [ "$BWD" = 1 ] && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/${TAG}bwd_fused_$tag -o pmc -- python $R/tools/run_train_leg.py --iters 6 > $O/pmc_bwd_$tag.log 2>&1
done
# 3. SQ counters of the default schedule and of the multi-launch schedule
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq/default_$tag -o pmc -- $B --steps 6 --warmup 2 --no-per-step-leg > $O/sq_default_$tag.log 2>&1
  CSPN_RESIDENT=off rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq/multi_$tag -o pmc -- $B --steps 6 --warmup 2 --no-per-step-leg > $O/sq_multi_$tag.log 2>&1
  [ "$BWD" = 1 ] && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq/bwd_$tag -o pmc -- python $R/tools/run_train_leg.py --iters 6 > $O/sq_bwd_$tag.log 2>&1
done
cd $R
export CSPN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null)
python tools/pmc_traffic.py $O/pmc $TAG > $O/traffic_$TAG.json
[ "$BWD" = 1 ] && python tools/pmc_traffic.py $O/pmc ${TAG}bwd > $O/traffic_${TAG}bwd.json
python tools/pmc_sq_summary.py $O/sq $O/sq_$TAG.json > $O/sq_$TAG.txt
for d in stats_default stats_s1 stats_multi stats_bwd; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep cspn $f) > $O/${d}_cspn.csv
done
ls $O; tail -3 $O/stats_default.log
