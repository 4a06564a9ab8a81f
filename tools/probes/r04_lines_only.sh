#!/bin/bash
# bench lines + PAC op table + scale-sweep dry run on the committed PMC traffic (no kernel of the bench workloads changed since)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${SESSION_TAG:-r04lines5}; mkdir -p $O; cd $R
bash tools/r04_bench_lines.sh > $O/bench_lines.log 2>&1; tail -16 $O/bench_lines.log
timeout 900 python tools/bench_pac_conv.py --json $O/pac_conv_unpool.json > $O/pac_conv_unpool.log 2>&1; cut -c1-150 $O/pac_conv_unpool.log
GPUS="1 2" BACKEND=gloo STEPS=20 WORKLOADS="nyu kitti pac5" OUT=$O/scale bash tools/scale_sweep.sh > $O/scale_sweep.txt 2>&1; tail -8 $O/scale_sweep.txt
