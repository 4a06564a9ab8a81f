#!/bin/bash
# Regenerate cspn_monodepth_amd/network/miopen_db/ (run on an MI355X; ~8.5 minutes): the reference's cudnn.benchmark search
# (main.py:37) for config 5's per-GPU shard, with MIOpen writing its user find-db / perf-db where we can pick them up.
#   gpurun --timeout 2400 -- 'bash tools/make_miopen_db.sh'   ->   gpurun_out/miopen_db/*.{ufdb,udb}.txt
set -eu
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/miopen_db
export MIOPEN_USER_DB_PATH=$R/gpurun_out/miopen_db
python $R/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline --conv-autotune on --conv-db off
ls -la $MIOPEN_USER_DB_PATH
echo "copy gpurun_out/miopen_db/*.udb.txt *.ufdb.txt into cspn_monodepth_amd/network/miopen_db/"
