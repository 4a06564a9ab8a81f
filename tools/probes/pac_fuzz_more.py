#!/usr/bin/env python3
"""Developer probe: the seeded geometry sweeps of tests/test_hip_pac_conv.py over seeds the test-suite does not use
(`python tools/probes/pac_fuzz_more.py 1000 1600`)."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_pac_conv as t            # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    for fn in (t.test_fuzz_geometry_vs_oracle, t.test_fuzz_tiled_geometry_both_kernels,
               t.test_fuzz_unit_stride_dilated_and_rectangular_windows, t.test_fuzz_stride2_register_kernels_and_generic):
        try:
            fn(seed)
        except AssertionError:
            bad += 1
            print("seed %d %s:" % (seed, fn.__name__))
            traceback.print_exc(limit=1)
print("seeds %d..%d x 4 sweeps: %d failing cases" % (lo, hi, bad))
