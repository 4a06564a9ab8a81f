#!/bin/bash
# tools/scale_sweep.sh — the north-star scaling table: N in {1,2,4,8} x {nyu, kitti, pac5, train} through the driver's own launch
# line (python -m torch.distributed.run ... bench.py --gpus N), then tools/scale_table.py prints absolute depth-maps/s, scaling
# efficiency and the fraction of N x the single-GPU HBM roofline of each workload.
#   bash tools/scale_sweep.sh                      # a real node: N = 1 2 4 8 over RCCL
#   GPUS="1 2" BACKEND=gloo STEPS=4 WORKLOADS="nyu kitti" bash tools/scale_sweep.sh     # dry run on one GPU (oversubscribed)
# Environment: GPUS (default "1 2 4 8"), WORKLOADS (default "nyu kitti pac5 train"), BACKEND (nccl), STEPS (100; train: 20),
# OUT (gpurun_out/scale), PORT (29800).  N is capped at the number of visible GPUs unless BACKEND=gloo.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
GPUS=${GPUS:-"1 2 4 8"}
WORKLOADS=${WORKLOADS:-"nyu kitti pac5 train"}
BACKEND=${BACKEND:-nccl}
STEPS=${STEPS:-100}
OUT=${OUT:-$R/gpurun_out/scale}
PORT=${PORT:-29800}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
mkdir -p "$OUT"
cd "$R"
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
for wl in $WORKLOADS; do
  for n in $GPUS; do
    if [ "$BACKEND" != "gloo" ] && [ "$n" -gt "$NDEV" ]; then echo "skip $wl N=$n: only $NDEV GPU(s) visible"; continue; fi
    steps=$STEPS; warm=10; extra="--no-cpu-baseline --no-train-leg --no-per-step-leg --cold-sets 0"
    if [ "$wl" = "train" ]; then steps=$(( STEPS < 20 ? STEPS : 20 )); warm=3; extra="--no-cpu-baseline"; fi
    f="$OUT/scale_${wl}_n${n}.json"
    if [ "$n" -eq 1 ]; then
      timeout 1200 python bench.py --gpus 1 --workload $wl --steps $steps --warmup $warm $extra 2>"$OUT/scale_${wl}_n${n}.err" | tail -1 > "$f"
    else
      PORT=$((PORT + 1))
      timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $n --backend $BACKEND --workload $wl --steps $steps --warmup $warm $extra 2>"$OUT/scale_${wl}_n${n}.err" | grep '^{' | tail -1 > "$f"
    fi
    [ -s "$f" ] || echo "FAILED $wl N=$n (see $OUT/scale_${wl}_n${n}.err)"
  done
done
python tools/scale_table.py "$OUT"
