#!/usr/bin/env python3
"""Golden vectors G11/G12 (round 2), made by IMPORTING the reference.  Build container only (needs /root/reference):

    python tests/golden/make_golden_r02.py

G11  CSPN_ours.AffinityPropagate with a MULTI-CHANNEL x [B,C>1,H,W] (the reference documents N,C,H,W,
     CSPN_ours.py:24-29; pac.conv2d broadcasts the shared kernel, pac.py:89-92): forward through the module itself,
     gradients by autograd through the reference's own differentiable branch (pac.conv2d(native_impl=True),
     pac.py:130-140) in fp64.
G12  gradient of the reference's nd2col (pac.py:35-70; differentiable through F.unfold / conv_transpose2d / F.pad):
     d(sum(cols * cot))/d(input) for plain, dilated and transposed geometry, and the gradients of
     pac.conv2d(native_impl=True) that flow THROUGH nd2col (already pinned by G9; re-used here as the same op).

While the reference is importable the numpy oracle additions (oracle/cspn_oracle.pac_*_multichannel,
oracle/pac_oracle.nd2col_backward) are checked against it; deviations go to golden_r02_manifest.json.
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
from network.libs.base import pac as ref_pac                # noqa: E402  (reference)
from oracle import cspn_oracle as orc                       # noqa: E402
from oracle import pac_oracle as porc                       # noqa: E402

torch.set_num_threads(4)
manifest = {"files": {}, "oracle_vs_reference": {}}


def t(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: v for k, v in arrs.items() if v is not None})
    manifest["files"][name] = {"bytes": os.path.getsize(path),
                               "arrays": {k: list(np.shape(v)) for k, v in arrs.items() if v is not None}}


def g11():
    B, H, W = 2, 14, 20
    for K, T, C in ((3, 6, 3), (5, 4, 2)):
        NT = K * K - 1
        gd = orc.hash_normal(110 + K, 1, (B, NT, H, W))
        x = orc.hash_uniform(110 + K, 2, (B, C, H, W), 0.0, 10.0)
        s = orc.hash_sparse(110 + K, 3, orc.hash_uniform(110 + K, 5, (B, 1, H, W), 0.0, 10.0), 0.08)
        cot = orc.hash_normal(110 + K, 4, (B, C, H, W))
        for sp in (None, s):
            tag = "g11_k%d_c%d_%s" % (K, C, "sp" if sp is not None else "nosp")
            with torch.no_grad():
                out = CSPN_ours.AffinityPropagate(T)(t(x), t(gd), sparse_depth=t(sp)).numpy()
            assert out.shape == x.shape
            xt = t(x).double().requires_grad_(True)
            gt = t(gd).double().requires_grad_(True)
            sm = torch.softmax(gt, dim=1)
            kern = torch.zeros(B, NT + 1, H, W, dtype=torch.float64)
            kern[:, :NT // 2] = sm[:, :NT // 2]
            kern[:, NT // 2 + 1:] = sm[:, NT // 2:]
            kern = kern.reshape(B, 1, K, K, H, W)
            cur = xt
            for _ in range(T):
                cur = ref_pac.conv2d(cur, kern, kernel_size=K, stride=1, padding=K // 2, dilation=1, native_impl=True)
                if sp is not None:
                    m = t(sp).double().sign()
                    cur = m * xt + (1 - m) * cur
            cur.backward(t(cot).double())
            assert np.abs(cur.detach().numpy() - out).max() < 1e-4
            save(tag, x=x, guided=gd, sparse=sp, out=out, cot=cot, grad_x=xt.grad.numpy(), grad_guided=gt.grad.numpy(),
                 T=np.int32(T), K=np.int32(K))
            o = orc.pac_forward_multichannel(x, gd, sp, T)
            ox, og = orc.pac_backward_multichannel(x, gd, sp, cot, T, np.float64)
            e = [float(np.abs(o - out).max()), float(np.abs(ox - xt.grad.numpy()).max()),
                 float(np.abs(og - gt.grad.numpy()).max())]
            manifest["oracle_vs_reference"][tag] = e
            assert e[0] < 2e-5 and e[1] < 1e-10 and e[2] < 1e-10, (tag, e)


def g12():
    x = orc.hash_normal(120, 1, (2, 3, 7, 9))
    cases = {"plain": dict(kernel_size=3, stride=1, padding=1, dilation=1),
             "stride2": dict(kernel_size=3, stride=2, padding=1, dilation=1),
             "dil": dict(kernel_size=(3, 2), stride=1, padding=(2, 0), dilation=(2, 3)),
             "k5": dict(kernel_size=5, stride=1, padding=2, dilation=1),
             "transposed": dict(kernel_size=3, stride=2, padding=1, output_padding=1, dilation=1, transposed=True),
             "transposed_dil": dict(kernel_size=3, stride=3, padding=2, output_padding=2, dilation=2, transposed=True)}
    for idx, (name, kw) in enumerate(cases.items()):
        xin = np.ascontiguousarray(x[:, :1] if kw.get("transposed") else x)     # pac.py:53: ones kernel [1,1,1,1] -> C = 1
        xt = t(xin).double().requires_grad_(True)
        cols = ref_pac.nd2col(xt, **kw)
        cot = orc.hash_normal(121 + idx, 2, tuple(cols.shape))
        (cols * t(cot).double()).sum().backward()
        full = dict(kernel_size=1, stride=1, padding=0, output_padding=0, dilation=1, transposed=False)
        full.update(kw)
        geom = np.array(sum((list(porc._pair(full[key])) for key in ("kernel_size", "stride", "padding", "dilation",
                                                                     "output_padding")), []) + [int(full["transposed"])],
                        np.int32)
        save("g12_nd2col_grad_" + name, x=xin, cot=cot, grad_x=xt.grad.numpy(), geom=geom)
        mine = porc.nd2col_backward(cot.astype(np.float64), xin.shape[-2:], **kw)
        e = float(np.abs(mine - xt.grad.numpy()).max())
        manifest["oracle_vs_reference"]["nd2col_grad_" + name] = e
        assert e < 1e-12, (name, e)


def g13():
    """Host model of config 5 (network/unet_cspn_nyu.py): (a) the reference's state_dict keys + shapes (checkpoint
    compatibility), (b) one decoder block of each kind on small tensors with its weights stored, (c) the full seeded,
    untrained resnet50 on one 228x304 RGB-D frame: the tensors handed to the CSPN module and the refined depth,
    sub-sampled (the weights are regenerated on the test side from the same seed and construction order)."""
    from network import unet_cspn_nyu as ref_net
    torch.manual_seed(0)
    net = ref_net.resnet50(pretrained=False).eval()
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, "g13_unet_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    manifest["g13_params"] = int(sum(p.numel() for p in net.parameters()))
    cap = {}
    net.post_process_layer.register_forward_hook(lambda m, i, o: cap.update(i=i, o=o))
    rgb = orc.hash_uniform(130, 1, (1, 3, 228, 304), 0.0, 1.0)
    dep = orc.hash_uniform(130, 2, (1, 1, 228, 304), 0.5, 10.0)
    sp = orc.hash_sparse(130, 3, dep, 500.0 / (228 * 304))
    x = np.concatenate([rgb, sp], 1)
    with torch.no_grad():
        out = net(t(x)).numpy()
    gdn, coarse, sparse = (v.numpy() for v in cap["i"])
    sub = 4
    save("g13_unet_full", seed=np.int32(0), out_sub=out[:, :, ::sub, ::sub], guidance_sub=gdn[:, :, ::sub, ::sub],
         coarse_sub=coarse[:, :, ::sub, ::sub], sub=np.int32(sub),
         moments=np.array([out.astype(np.float64).sum(), (out.astype(np.float64) ** 2).sum(),
                           gdn.astype(np.float64).sum(), (gdn.astype(np.float64) ** 2).sum()]))
    assert np.array_equal(sparse, sp)
    # (b) decoder blocks, train-mode BN (batch statistics) so the BN arithmetic is exercised
    torch.manual_seed(1)
    blocks = {"gudi": (ref_net.Gudi_UpProj_Block(6, 4, 7, 9), False), "cat": (ref_net.Gudi_UpProj_Block_Cat(6, 4, 7, 9), True),
              "last": (ref_net.Simple_Gudi_UpConv_Block_Last_Layer(6, 3, 8, 10), False)}
    for name, (blk, has_side) in blocks.items():
        xin = orc.hash_normal(131, len(name), (2, 6, 4, 5))
        side = orc.hash_normal(132, len(name), (2, 4, 7, 9)) if has_side else None
        blk.train()
        with torch.no_grad():
            y = (blk(t(xin), t(side)) if has_side else blk(t(xin))).numpy()
        sd = {"sd_" + k: v.numpy() for k, v in blk.state_dict().items() if "running" not in k and "num_batches" not in k}
        save("g13_block_" + name, x=xin, side=side, out=y, **sd)


if __name__ == "__main__":
    g11()
    g12()
    g13()
    manifest["torch"] = torch.__version__
    manifest["numpy"] = np.__version__
    with open(os.path.join(HERE, "golden_r02_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("total fixture bytes", sum(v["bytes"] for v in manifest["files"].values()))
