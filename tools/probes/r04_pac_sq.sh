#!/bin/bash
# SQ counters of the C = 64 shared 5x5 rows (forward, dL/dinput, dL/dkernel): what are the LDS-tiled kernels waiting for?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pacsq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq/default_$tag -o pmc -- python $R/tools/bench_pac_conv.py --only c64_k5 --reps 3 > $O/sq_$tag.log 2>&1
done
cd $R
python tools/pmc_sq_summary.py $O/sq $O/sq.json > $O/sq.txt 2>&1; grep -A14 "pac_conv2d" $O/sq.txt | cut -c1-200 | head -80
