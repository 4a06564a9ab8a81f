#!/bin/bash
# Round 5: the K = 5 reverse sweep as ONE launch (cspnk_resident<TRANS>: rounds x images_per_launch x tiles workgroups) against one launch per
# round (-DCSPN_KT_ONE_LAUNCH=0), same box, interleaved: the parity tests of the K x K training path on the in-tree library, then per variant the
# kernel statistics of the fp16 training leg (config 3's shape, plain and sparse) and the wall time of the step; a checksum of both gradients
# (the variants must agree bit for bit).
# usage (gpurun): bash tools/probes/r05_sweep_one_launch.sh <variant tags under _ab/ ...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05sweep
mkdir -p $O
cd $R
python -m pytest tests/test_hip_kres.py tests/test_hip_pac_conv.py tests/test_hip_production.py tests/test_hip_resident.py -x -q -m gpu -k "trans or backward or train or grad or hist or config3" 2>&1 | tail -3 > $O/pytest_tail.txt
cat $O/pytest_tail.txt
for r in 1 2; do for v in "$@"; do
  export CSPN_HIP_LIB=$R/_ab/lib_$v.so
  for sp in "" "--sparse"; do
    n=stats_${v}${sp:+_sparse}_$r
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o bwd -- python tools/run_train_leg.py --K 5 --dtype f16 --state input --iters 40 $sp > $O/$n.log 2>&1
    f=$(find $O/$n -name "*kernel_stats.csv" | head -1)
    echo "== $n"; head -5 "$f" | cut -d, -f1-4 | cut -c1-200
    cp "$f" $O/$n.csv; rm -rf $O/$n
  done
  python - <<PY
import sys, torch
sys.path.insert(0, ".")
import cspn_monodepth_amd as pkg
torch.manual_seed(0)
B, H, W, T = 24, 228, 304, 12
g = torch.randn(B, 24, H, W, device="cuda").half().requires_grad_(True)
d = (torch.rand(B, 1, H, W, device="cuda") * 10).half().requires_grad_(True)
cot = torch.randn(B, 1, H, W, device="cuda").half()
m = pkg.CSPN_ours.AffinityPropagate(T, state_dtype=None)
for sp in (None, (d.detach() * (torch.rand_like(d) < 0.007))):
    def step():
        g.grad = None; d.grad = None
        m(d, g, sparse_depth=sp).backward(cot)
    for _ in range(20): step()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): step()
        e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 1000 / 50)
    print("$v round $r %s: %.1f us per training step; grad sums %.6f %.6f" % ("sparse" if sp is not None else "plain", best, float(g.grad.double().sum()), float(d.grad.double().sum())))
PY
done; done 2>&1 | tee $O/ab.txt
