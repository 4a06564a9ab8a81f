#!/usr/bin/env python3
"""Where does the metrics kernel's time go?  (developer micro-benchmark)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_monodepth_amd as pkg
from tools.tune import timed
d = torch.rand(24, 1, 228, 304, device="cuda") * 10 + 0.1
t = d + 0.1
acc = pkg.evaluation.new_accumulator("cuda")
print("metric_sums: %.1f us" % timed(lambda: pkg.evaluation.metric_sums(d, t, out=acc), 50))
print("torch sum  : %.1f us" % timed(lambda: (d - t).abs().sum(), 50))
