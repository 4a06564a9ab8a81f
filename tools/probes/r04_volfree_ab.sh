#!/bin/bash
# alternating same-box A/B of the training-shaped step with (1) and without (0) the tap volume: wall time per forward + backward
for k in 1 2 3; do for v in 0 1; do
echo "== CSPN_TRAIN_VOLUME=$v run $k"; CSPN_TRAIN_VOLUME=$v python tools/probes/bench_backward.py --reps 40 2>&1 | grep -v amdgpu.ids
done; done
echo "== host split"; for v in 0 1; do CSPN_TRAIN_VOLUME=$v python tools/probes/host_split_training.py 2>&1 | tail -12; done
