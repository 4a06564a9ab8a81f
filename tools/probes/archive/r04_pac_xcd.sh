#!/bin/bash
# XCD-aware chunk mapping of the LDS-tiled PAC kernels: the whole PAC test file + the unpool tests, then the op table
O=gpurun_out/pacxcd; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_pac_conv.py tests/test_pac_conv_oracle.py tests/test_unpool.py tests/test_hip_fuzz.py -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python tools/bench_pac_conv.py --only c --json $O/pac_conv.json > $O/bench.log 2>&1; cut -c1-150 $O/bench.log
