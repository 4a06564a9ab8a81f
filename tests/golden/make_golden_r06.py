#!/usr/bin/env python3
"""Golden vector G15 (round 6), made by IMPORTING the reference.  Build container only (needs /root/reference):

    python tests/golden/make_golden_r06.py

G15  an external anchor for the fp16 tolerances of BASELINE config 3 ("fp16 tolerances are still the builder's
     own").  One full NYU frame (228 x 304, K = 5, T = 12), fp16-rounded inputs from the hash generator, through the reference's
     CSPN_ours.AffinityPropagate (CSPN_ours.py:24-54) twice:
       * `out_taps16`: half inputs under the default dtype float32 — the reference's own promotion rules give fp16 softmax taps and an
         fp32 state (its `kernel = torch.zeros(...)` is fp32, CSPN_ours.py:37), and
       * `out_half`: the same call with torch.set_default_dtype(float16) — what the reference computes when it runs in half precision
         end to end (taps, state and every step's accumulation rounded to half), i.e. config 3 as the reference would run it.
     The fixture stores both outputs; the test holds this package's fp16 paths to the distance the reference's OWN half-precision
     run keeps from the fp32 result on the same inputs.
"""
import json
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CSPN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
_stub = types.ModuleType("torch._thnn")          # pac.py:20 imports torch._thnn (removed in torch>=1.0)
_stub.type2backend = defaultdict(lambda: None)
sys.modules.setdefault("torch._thnn", _stub)

from network.libs.post_process import CSPN_ours             # noqa: E402  (reference)
from oracle import cspn_oracle as orc                       # noqa: E402

torch.set_num_threads(4)
B, H, W, K, T, SEED = 1, 228, 304, 5, 12, 45
gd = orc.hash_normal(SEED, 1, (B, K * K - 1, H, W)).astype(np.float16)
x = orc.hash_uniform(SEED, 2, (B, 1, H, W), 0.0, 10.0).astype(np.float16)
sp = orc.hash_sparse(SEED, 3, x.astype(np.float32), 500.0 / (H * W)).astype(np.float16)
m = CSPN_ours.AffinityPropagate(T)
res = {}
for tag, spd in (("nosp", None), ("sp", sp)):
    outs = {}
    for name, dd in (("taps16", torch.float32), ("half", torch.float16)):
        torch.set_default_dtype(dd)
        with torch.no_grad():
            o = m(torch.from_numpy(x), torch.from_numpy(gd), None if spd is None else torch.from_numpy(spd))
        torch.set_default_dtype(torch.float32)
        outs[name] = o
    assert outs["taps16"].dtype == torch.float32 and outs["half"].dtype == torch.float16
    want = orc.pac_forward(x.astype(np.float32), gd.astype(np.float32), None if spd is None else spd.astype(np.float32), T)
    scale = float(np.abs(want).max())
    e = {}
    for name in outs:
        o = outs[name].float().numpy()
        e[name] = {"max_over_scale": float(np.abs(o - want).max() / scale), "rmse_over_scale": float(np.sqrt(((o - want) ** 2).mean()) / scale)}
    res[tag] = e
    np.savez_compressed(os.path.join(HERE, "g15_k5_t12_fp16_frame_%s.npz" % tag), seed=np.int32(SEED), T=np.int32(T), K=np.int32(K),
                        shape=np.array([B, H, W], np.int32), sparse_rate=np.float64(500.0 / (H * W)),
                        out_taps16=outs["taps16"].numpy().astype(np.float32), out_half=outs["half"].numpy(),
                        ref_err_taps16=np.array([e["taps16"]["max_over_scale"], e["taps16"]["rmse_over_scale"]]),
                        ref_err_half=np.array([e["half"]["max_over_scale"], e["half"]["rmse_over_scale"]]))
json.dump({"g15_reference_error_vs_fp32_oracle_on_fp16_inputs": res, "torch": torch.__version__},
          open(os.path.join(HERE, "golden_r06_manifest.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
