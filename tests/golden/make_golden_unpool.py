#!/usr/bin/env python3
"""Golden vectors G10 for the zero-insertion un-pooling, made by IMPORTING the reference.

    python tests/golden/make_golden_unpool.py        (build container only: needs /root/reference)

Captures MyBlock._up_pooling (network/unet_ours.py:138-150) and Simple_Gudi_UpConv_Block_Last_Layer._up_pooling
(network/unet_cspn_nyu.py:202-213) outputs and their autograd gradients, and checks oracle/pac_oracle.up_pooling.
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
_stub = types.ModuleType("torch._thnn")
_stub.type2backend = defaultdict(lambda: None)
sys.modules.setdefault("torch._thnn", _stub)

from network import unet_cspn_nyu, unet_ours                # noqa: E402  (reference)
from oracle import cspn_oracle as orc                       # noqa: E402
from oracle import pac_oracle as porc                       # noqa: E402

manifest = {"files": {}, "checks": {}}
# name: (B, C, H, W, scale, oheight, owidth)
CASES = {"even": (2, 3, 5, 6, 2, 10, 12), "crop_odd": (1, 4, 8, 10, 2, 15, 19), "quarter_frame": (1, 1, 57, 76, 2, 114, 152),
         "scale3_crop": (1, 2, 4, 5, 3, 11, 13), "one_px": (1, 1, 1, 1, 2, 1, 1)}


def main():
    for idx, (name, (B, C, H, W, s, oh, ow)) in enumerate(CASES.items()):
        x = orc.hash_normal(1000 + idx, 1, (B, C, H, W))
        cot = orc.hash_normal(1000 + idx, 2, (B, C, oh, ow))
        xt = torch.from_numpy(x).requires_grad_(True)
        y = unet_ours.MyBlock(oh, ow)._up_pooling(xt, s)
        y.backward(torch.from_numpy(cot))
        arrs = dict(x=x, cot=cot, out=y.detach().numpy(), grad_x=xt.grad.numpy(), geom=np.array([s, oh, ow], np.int32))
        if s == 2:      # the mask-loop variant (scale 2 only: its loop steps by 2)
            with torch.no_grad():
                y2 = unet_cspn_nyu.Simple_Gudi_UpConv_Block_Last_Layer(1, 1, oh, ow)._up_pooling(torch.from_numpy(x), 2)
            assert np.array_equal(y2.numpy(), arrs["out"]), name
            manifest["checks"][name + "_variants_identical"] = True
        path = os.path.join(HERE, "g10_unpool_" + name + ".npz")
        np.savez_compressed(path, **arrs)
        manifest["files"]["g10_unpool_" + name] = os.path.getsize(path)
        assert np.array_equal(porc.up_pooling(x, s, oh, ow), arrs["out"]), name
        assert np.array_equal(porc.up_pooling_backward(cot, (H, W), s), arrs["grad_x"]), name
        manifest["checks"][name + "_oracle"] = "bit-identical"
    manifest["torch"] = torch.__version__
    with open(os.path.join(HERE, "golden_unpool_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("total bytes", sum(manifest["files"].values()))


if __name__ == "__main__":
    main()
