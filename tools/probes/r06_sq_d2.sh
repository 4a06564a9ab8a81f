#!/bin/bash
# round 6: SQ counters of config 3's kernel (cspnk_d2) for library variants under _ab/ — LDS bank conflicts before / after the DPP halo
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  B="python $R/bench.py --workload pac5 --no-cpu-baseline --no-train-leg --cold-sets 0 --prewarm-s 0 --no-per-step-leg --no-sparse-leg --no-stock-ops-leg --steps 6 --warmup 2"
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM"; do
    tag=$(echo $grp | cut -d' ' -f1)
    CSPN_HIP_LIB=$R/_ab/lib_$v.so rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq_$v/$tag -o pmc -- $B > $O/sq_${v}_$tag.log 2>&1
  done
  python $R/tools/pmc_sq_summary.py $O/sq_$v $O/sq_pac5_$v.json > $O/sq_pac5_$v.txt
  rm -rf $O/sq_$v
done
tail -30 $O/sq_pac5_*.txt
