#!/bin/bash
# round 6: the 8-filter affinity head against the reference's 12: config 5's step at B = 3 and an inference-only UNet + CSPN step at B = 24,
# two alternating rounds on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06h
for r in 1 2; do for c in 12 8; do
  python bench.py --workload train --steps 30 --warmup 8 --affinity-channels $c --infer-batch 24 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('round $r affinity_channels %d: train step %.2f ms (B=3), CSPN fwd %.1f us bwd %.1f us; inference step B=24 %.2f ms' % (d['config']['affinity_channels'], d['ms_per_step'], d['cspn_module']['forward_us_p50'], d['cspn_module']['backward_us_p50'], d['inference_step']['ms_per_step']))
"
done; done | tee gpurun_out/r06h/head_ab.txt
