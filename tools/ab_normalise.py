#!/usr/bin/env python3
"""A/B of the weight normalisation arithmetic (developer tool): error vs the goldens and forward time.

Build the alternative library next to the default one and run the tool once per build (the flags must be repeated on
the GPU box, or the content hash differs and the library is rebuilt with the default flags):

    CSPN_HIP_LIB=$PWD/_ab/libcspn_ieee.so CSPN_HIPCC_FLAGS=-DCSPN_IEEE_NORMALISE python -c \
        "from cspn_monodepth_amd import _lib; _lib.build(force=True)"
    gpurun -- 'python tools/ab_normalise.py; CSPN_HIPCC_FLAGS=-DCSPN_IEEE_NORMALISE \
        CSPN_HIP_LIB=$GRAFT_REPO_ROOT/_ab/libcspn_ieee.so python tools/ab_normalise.py'"""
import os, sys, glob
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
from tools.tune import timed
from bench import WORKLOADS, make_inputs
from conftest import load_golden, golden_names, rel_err, rmse
from oracle import c_oracle, cspn_oracle as orc
c_oracle.build()
from cspn_monodepth_amd import _lib as _L
print("library:", _L.SO_PATH, os.path.getsize(_L.SO_PATH))
worst = 0.0
for n in golden_names("g1_") + golden_names("g2_") + golden_names("g8_"):
    z = load_golden(n)
    if "guidance" in z: g, d, s = z["guidance"], z["blur"], z.get("sparse")
    else: g, d, s = (z["guidance_f16"].astype(np.float32), z["blur_f16"].astype(np.float32), z["sparse_f16"].astype(np.float32))
    T = int(z["T"])
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with torch.no_grad():
        out = pkg.CSPN_new.AffinityPropagate(T, 3)(dev(g), dev(d), dev(s)).cpu().numpy()
    worst = max(worst, rel_err(out, z["out"]))
print("worst rel err over small goldens: %.3e" % worst)
for seed, (B, H, W) in enumerate(((4, 228, 304), (1, 352, 1216))):
    for sp in (None, 500):
        g, d, s = orc.synthetic_inputs(seed=seed, B=B, H=H, W=W, C=12, sparse_samples=sp)
        want = c_oracle.cspn3_forward(g, d, s, 24)
        with torch.no_grad():
            out = pkg.CSPN_new.AffinityPropagate(24, 3)(torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda(),
                                                       None if s is None else torch.from_numpy(s).cuda()).cpu().numpy()
        print("full frame %dx%dx%d sparse=%s: rel %.3e rmse %.3e" % (B, H, W, sp, rel_err(out, want), rmse(out, want)))
wl = dict(WORKLOADS["nyu"])
g, d, s, _ = make_inputs(wl, 24, torch.device("cuda", 0), 1, False)
m = pkg.CSPN_new.AffinityPropagate(24, 3)
with torch.no_grad():
    print("forward %.1f us" % timed(lambda: m(g, d, None), 50))
    print("prepare %.1f us" % timed(lambda: F.cspn3_prepare(g), 50))
