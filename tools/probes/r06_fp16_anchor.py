#!/usr/bin/env python3
"""round 6: this package's fp16 paths of config 3 against the distance the REFERENCE's own half-precision run keeps from the fp32 result
(golden G15, tests/golden/make_golden_r06.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_monodepth_amd as pkg
from oracle import c_oracle
for tag in ("nosp", "sp"):
    z = np.load("tests/golden/g15_k5_t12_fp16_frame_%s.npz" % tag)
    B, H, W = (int(v) for v in z["shape"]); T, K, seed = int(z["T"]), int(z["K"]), int(z["seed"])
    gd = c_oracle.hash_normal(seed, 1, (B, K * K - 1, H, W)).astype(np.float16)
    x = c_oracle.hash_uniform(seed, 2, (B, 1, H, W), 0.0, 10.0).astype(np.float16)
    sp = c_oracle.hash_sparse(seed, 3, x.astype(np.float32), float(z["sparse_rate"])).astype(np.float16) if tag == "sp" else None
    want = c_oracle.pac_forward(x.astype(np.float32), gd.astype(np.float32), None if sp is None else sp.astype(np.float32), T)
    scale = float(np.abs(want).max())
    d = lambda a: None if a is None else torch.from_numpy(a).cuda()
    for B_rep in (1, 24):
        for state in ("reference", None):
            with torch.no_grad():
                o = pkg.CSPN_ours.AffinityPropagate(T, state_dtype=state)(d(x).repeat(B_rep, 1, 1, 1), d(gd).repeat(B_rep, 1, 1, 1),
                                                                         sparse_depth=None if sp is None else d(sp).repeat(B_rep, 1, 1, 1))
            o = o.float().cpu().numpy()[:1]
            ref = z["out_taps16"] if state == "reference" else z["out_half"].astype(np.float32)
            rerr = z["ref_err_taps16"] if state == "reference" else z["ref_err_half"]
            print("%s B=%d state=%s: ours vs oracle max %.3e rmse %.3e | reference's own %.3e %.3e | ours vs the reference's output max %.3e" % (
                tag, B_rep, state, np.abs(o - want).max() / scale, np.sqrt(((o - want) ** 2).mean()) / scale, rerr[0], rerr[1],
                np.abs(o - ref).max() / scale))
