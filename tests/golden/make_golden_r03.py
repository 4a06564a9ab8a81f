#!/usr/bin/env python3
"""Golden vector G14 (round 3), made by IMPORTING the reference.  Build container only (needs /root/reference):

    python tests/golden/make_golden_r03.py

G14  the tail of the model `get_model` actually returns (network/__init__.py:19-20 -> network/unet_ours.py), chained as its
     forward does (unet_ours.py:325-333): decoder features -> Gudi_UpProj_Block_Cat (MyBlock._up_pooling with a crop,
     :138-150, :226-250) -> the two Simple_Gudi_UpConv_Block_Last_Layer heads (blur depth, 8-channel guidance, :194-203)
     -> CSPN_ours.AffinityPropagate(prop_time)(blur_depth, guidance, sparse_depth=...) (:333).  The reference classes are
     instantiated as they are, seeded, in fp64; forward through the modules themselves, gradients by autograd with the
     reference's own differentiable branch of pac.conv2d (native_impl=True, pac.py:130-140 — Conv2dFn.backward needs the
     removed torch._thnn).  Stored: inputs, every parameter / buffer of the three blocks (state_dict order), the
     intermediate maps, the refined depth, and the gradients w.r.t. the features, the side input and two of the weights.
"""
import functools
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

from network import unet_ours as ref                         # noqa: E402  (reference)
from network.libs.base import pac as ref_pac                 # noqa: E402  (reference)
from oracle import cspn_oracle as orc                        # noqa: E402

torch.set_num_threads(4)


def main():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(1403)
    B, C, h, w = 2, 12, 10, 12
    oh1, ow1 = 19, 24                      # un-pooled 20 x 24, cropped to 19 rows (unet_ours.py:147-148)
    oh2, ow2 = 38, 48
    T = 6
    cat = ref.Gudi_UpProj_Block_Cat(C, 8, oh1, ow1)
    head_d = ref.Simple_Gudi_UpConv_Block_Last_Layer(8, 1, oh2, ow2)
    head_g = ref.Simple_Gudi_UpConv_Block_Last_Layer(8, 8, oh2, ow2)
    cspn = ref.post_process.AffinityPropagate(prop_time=T)
    for m in (cat, head_d, head_g):
        m.double()
    feat = torch.from_numpy(orc.hash_normal(1401, 1, (B, C, h, w)).astype(np.float64)).requires_grad_(True)
    side = torch.from_numpy(orc.hash_normal(1401, 2, (B, 8, oh1, ow1)).astype(np.float64)).requires_grad_(True)
    depth = orc.hash_uniform(1401, 3, (B, 1, oh2, ow2), 0.5, 10.0)
    sparse = torch.from_numpy(orc.hash_sparse(1401, 4, depth, 0.03).astype(np.float64))
    cot = torch.from_numpy(orc.hash_normal(1401, 5, (B, 1, oh2, ow2)).astype(np.float64))

    def forward(native):
        x = cat(feat, side)                                   # train mode: batch statistics (deterministic)
        blur = head_d(x)
        guid = head_g(x)
        if native:
            orig = ref_pac.conv2d
            ref_pac.conv2d = functools.partial(orig, native_impl=True)
            try:
                out = cspn(blur, guid, sparse_depth=sparse)
            finally:
                ref_pac.conv2d = orig
        else:
            out = cspn(blur, guid, sparse_depth=sparse)
        return x, blur, guid, out

    with torch.no_grad():
        _, _, _, out_module = forward(False)                 # the module exactly as shipped (Conv2dFn.forward)
    # the un-pooling of the reference on its own (MyBlock._up_pooling: grouped conv_transpose2d + crop)
    with torch.no_grad():
        up = cat._up_pooling(feat.detach(), 2)
    for m in (cat, head_d, head_g):                          # fresh running statistics for the recorded pass
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.reset_running_stats()
    x, blur, guid, out = forward(True)
    assert float((out.detach() - out_module).abs().max()) == 0.0, "native_impl branch differs from Conv2dFn.forward"
    (out * cot).sum().backward()
    arrs = dict(feat=feat.detach().numpy(), side=side.detach().numpy(), sparse=sparse.numpy(), cot=cot.numpy(), T=np.int64(T),
                sizes=np.array([oh1, ow1, oh2, ow2], np.int64), up=up.numpy(), x=x.detach().numpy(), blur=blur.detach().numpy(),
                guidance=guid.detach().numpy(), out=out.detach().numpy(), grad_feat=feat.grad.numpy(), grad_side=side.grad.numpy(),
                grad_head_g_weight=head_g.conv1.weight.grad.numpy(), grad_cat_conv1_weight=cat.conv1.weight.grad.numpy())
    for name, m in (("cat", cat), ("head_d", head_d), ("head_g", head_g)):
        for k, v in m.state_dict().items():
            if "num_batches_tracked" not in k:
                arrs["%s.%s" % (name, k)] = v.detach().numpy()
    path = os.path.join(HERE, "g14_unet_ours_tail.npz")
    np.savez_compressed(path, **arrs)
    json.dump({"file": "g14_unet_ours_tail.npz", "bytes": os.path.getsize(path), "arrays": {k: list(np.shape(v)) for k, v in arrs.items()},
               "reference": "network/unet_ours.py:131-250, :325-333; network/libs/post_process/CSPN_ours.py:24-54",
               "native_impl_vs_module_max_abs": 0.0},
              open(os.path.join(HERE, "golden_r03_manifest.json"), "w"), indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
