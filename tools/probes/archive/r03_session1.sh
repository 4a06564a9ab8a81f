#!/bin/bash
# Round-3 GPU session 1 (run through gpurun): new safety / DDP / parity tests first, then the whole GPU suite, bench lines
# with the repaired traffic accounting + S=1 cache-cold leg, and the K=5 training-leg profile (VERDICT r2 missing #5).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03s1
mkdir -p $O
cd $R
timeout 900 python -m pytest -q -x tests/test_hip_resident.py -k "timeout or many_phases" tests/test_hip_pac_conv.py::test_fp16_eight_pixel_kernel_several_channels_per_workgroup > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log
tail -5 $O/pytest_new.log
timeout 1700 python -m pytest -q -x tests/test_distributed_gpu.py -k "ddp or bench_train" > $O/pytest_ddp.log 2>&1; echo "rc=$?" >> $O/pytest_ddp.log
tail -15 $O/pytest_ddp.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_driver.log 2>$O/bench_default_driver.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default.log 2>&1
timeout 300 python bench.py --workload pac5 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_pac5.log 2>&1
timeout 300 python bench.py --workload kitti --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_kitti.log 2>&1
for f in $O/bench_*.log; do echo "== $f"; tail -1 $f | cut -c1-400; done
# K=5 training leg: kernel stats + HBM traffic
cd /tmp && export TMPDIR=/tmp
for cfg in "f16" "f32"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd_pac5_$cfg -o bwd -- python $R/tools/run_train_leg.py --K 5 --dtype $cfg --iters 30 > $O/stats_bwd_pac5_$cfg.log 2>&1
  f=$(find $O/stats_bwd_pac5_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep -i "cspn\|elementwise\|copy" $f) > $O/stats_bwd_pac5_${cfg}_cspn.csv
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/pac5bwd_fused_$ctr -o pmc -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --iters 6 > $O/pmc_pac5bwd_$ctr.log 2>&1
done
cd $R
export CSPN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null)
python tools/pmc_traffic.py $O/pmc pac5bwd > $O/traffic_pac5bwd.json
cat $O/stats_bwd_pac5_f16_cspn.csv | cut -c1-200
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_distributed_gpu.py::test_ddp_training_step_two_ranks --deselect tests/test_distributed_gpu.py::test_bench_train_two_ranks_on_one_gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
