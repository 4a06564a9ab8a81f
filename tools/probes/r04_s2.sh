#!/bin/bash
# Stride-2 PAC register kernels: parity (new fuzz + full-frame tests, then the whole PAC file), then the roofline rows.
set -x
O=gpurun_out/s2
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_pac_conv.py -q -x -k "stride2 or unfold_formulation" > $O/pytest_s2.log 2>&1
tail -5 $O/pytest_s2.log
timeout 1200 python -m pytest tests/test_hip_pac_conv.py tests/test_pac_conv_oracle.py -q > $O/pytest_pac.log 2>&1
tail -5 $O/pytest_pac.log
timeout 600 python tools/bench_pac_conv.py --only stride2 --json $O/pac_s2.json > $O/bench_s2.log 2>&1
cat $O/bench_s2.log
CSPN_PAC_SCALAR=1 timeout 600 python tools/bench_pac_conv.py --only stride2 > $O/bench_s2_generic.log 2>&1
cat $O/bench_s2_generic.log
