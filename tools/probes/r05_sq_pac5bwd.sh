#!/bin/bash
# SQ / TCC counters of the K = 5 fp16 training leg's kernels (cspnk_d2 with history, cspnk_resident<TRANS>, cspn_grad_tail5)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05sq_pac5bwd; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o pmc -- python $R/tools/run_train_leg.py --K 5 --dtype f16 --state input --iters 6 > $O/g$i.log 2>&1
done
cd $R; python tools/pmc_sq_summary.py $O $O/sq_pac5bwd.json > $O/sq_pac5bwd.txt; python - <<PY
import json
j=json.load(open("$O/sq_pac5bwd.json"))["per_kernel"]
for k,v in j.items():
    print(k); print("   ", {n: (round(x,4) if x<10 else round(x)) for n,x in v.items()})
PY
