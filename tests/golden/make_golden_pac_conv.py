#!/usr/bin/env python3
"""Golden vectors G9 for the general pixel-adaptive conv op, made by IMPORTING the reference.

Runs only in the build container (needs /root/reference).  Re-run:

    python tests/golden/make_golden_pac_conv.py

For every case: inputs, the output of the reference's ``pac.conv2d`` through BOTH of its branches
(``Conv2dFn.forward`` pac.py:75-94 and ``native_impl=True`` pac.py:130-140 — they must agree), and the
gradients autograd gives through the ``native_impl=True`` branch (``Conv2dFn.backward`` needs the THNN backend
torch removed).  ``nd2col`` (pac.py:35-70) is captured for plain and transposed geometry.  While the reference is
importable the numpy oracle (oracle/pac_oracle.py) is checked against it and the deviations recorded in
golden_pac_conv_manifest.json.
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

# pac.py:20 imports torch._thnn (removed in torch>=1.0); forward only needs the name.
_stub = types.ModuleType("torch._thnn")
_stub.type2backend = defaultdict(lambda: None)
sys.modules.setdefault("torch._thnn", _stub)

from network.libs.base import pac as ref_pac                # noqa: E402  (reference)
from oracle import cspn_oracle as orc                       # noqa: E402
from oracle import pac_oracle as porc                       # noqa: E402

torch.set_num_threads(4)
manifest = {"files": {}, "oracle_vs_reference": {}}

# name: (B, C, kernel_ch, H, W, kernel_size, stride, padding, dilation)
CASES = {
    "k3_same_shared":   (2, 3, 1, 11, 13, 3, 1, 1, 1),
    "k3_same_perch":    (2, 3, 3, 11, 13, 3, 1, 1, 1),
    "k5_same_c1":       (1, 1, 1, 12, 16, 5, 1, 2, 1),       # the geometry CSPN_ours.py:52 uses
    "k7_same_c2":       (1, 2, 1, 9, 12, 7, 1, 3, 1),
    "k1":               (1, 4, 4, 5, 8, 1, 1, 0, 1),
    "k3_valid":         (1, 2, 1, 10, 12, 3, 1, 0, 1),
    "k3_stride2":       (2, 2, 1, 11, 14, 3, 2, 1, 1),
    "k3_dil2":          (1, 3, 3, 12, 12, 3, 1, 2, 2),
    "k3_dil3_stride2":  (1, 2, 1, 15, 17, 3, 2, 3, 3),
    "rect":             (1, 2, 2, 10, 15, (3, 5), (1, 2), (0, 3), (2, 1)),
    "k5_wide_c5":       (1, 5, 1, 6, 40, 5, 1, 2, 1),
    "k3_pad_gt":        (1, 1, 1, 6, 7, 3, 1, 2, 1),         # padding larger than K//2: output bigger than input
}


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    manifest["files"][name] = {"bytes": os.path.getsize(path), "arrays": {k: list(np.shape(v)) for k, v in arrs.items()}}


def main():
    for idx, (name, (B, C, CK, H, W, k, s, p, d)) in enumerate(CASES.items()):
        Ho, Wo = porc.out_size((H, W), k, s, p, d)
        kh, kw = porc._pair(k)
        x = orc.hash_normal(900 + idx, 1, (B, C, H, W))
        kern = orc.hash_normal(900 + idx, 2, (B, CK, kh, kw, Ho, Wo))
        cot = orc.hash_normal(900 + idx, 3, (B, C, Ho, Wo))
        with torch.no_grad():
            out_fn = ref_pac.conv2d(torch.from_numpy(x), torch.from_numpy(kern), k, s, p, d).numpy()
        grads = {}
        for dt, nm in ((torch.float64, "f64"), (torch.float32, "f32")):
            xt = torch.from_numpy(x).to(dt).requires_grad_(True)
            kt = torch.from_numpy(kern).to(dt).requires_grad_(True)
            out_n = ref_pac.conv2d(xt, kt, k, s, p, d, native_impl=True)
            out_n.backward(torch.from_numpy(cot).to(dt))
            grads["grad_input_" + nm] = xt.grad.numpy()
            grads["grad_kernel_" + nm] = kt.grad.numpy()
            if dt == torch.float32:
                assert np.array_equal(out_n.detach().numpy().shape, out_fn.shape)
                manifest["oracle_vs_reference"].setdefault("branches_max_abs", {})[name] = float(
                    np.abs(out_n.detach().numpy() - out_fn).max())
        geom = np.array(list(porc._pair(k)) + list(porc._pair(s)) + list(porc._pair(p)) + list(porc._pair(d)), np.int32)
        save("g9_" + name, x=x, kernel=kern, cot=cot, out=out_fn, geom=geom, **grads)
        o = porc.pac_conv2d_forward(x, kern, k, s, p, d)
        gi, gk = porc.pac_conv2d_backward(x, kern, cot, k, s, p, d)
        e = [float(np.abs(o - out_fn).max()), float(np.abs(gi - grads["grad_input_f64"]).max()),
             float(np.abs(gk - grads["grad_kernel_f64"]).max())]
        manifest["oracle_vs_reference"][name] = e
        assert e[0] < 2e-5 and e[1] < 1e-12 and e[2] < 1e-12, (name, e)
    # fp16 forward (the reference multiplies and sums in half)
    x = orc.hash_normal(950, 1, (1, 2, 10, 12)).astype(np.float16)
    kern = (orc.hash_normal(950, 2, (1, 1, 3, 3, 10, 12)) * np.float32(0.3)).astype(np.float16)
    with torch.no_grad():
        out = ref_pac.conv2d(torch.from_numpy(x), torch.from_numpy(kern), 3, 1, 1, 1).numpy()
    save("g9_fp16", x=x, kernel=kern, out=out, geom=np.array([3, 3, 1, 1, 1, 1, 1, 1], np.int32))
    manifest["g9_fp16_out_dtype"] = str(out.dtype)
    # nd2col, plain and transposed (pac.py:51-58)
    x = orc.hash_normal(960, 1, (2, 2, 6, 7))
    for name, kw in {"plain": dict(kernel_size=3, stride=2, padding=1, dilation=1),
                     "dil": dict(kernel_size=(3, 2), stride=1, padding=(2, 0), dilation=(2, 3)),
                     "transposed": dict(kernel_size=3, stride=2, padding=1, output_padding=1, dilation=1, transposed=True),
                     "transposed_k4": dict(kernel_size=4, stride=2, padding=1, output_padding=0, dilation=1, transposed=True),
                     "transposed_dil": dict(kernel_size=3, stride=3, padding=2, output_padding=2, dilation=2, transposed=True),
                     }.items():
        xin = x[:, :1] if kw.get("transposed") else x      # pac.py:53-55: the ones kernel is [1,1,1,1] -> C must be 1
        with torch.no_grad():
            cols = ref_pac.nd2col(torch.from_numpy(np.ascontiguousarray(xin)), **kw).numpy()
        full = dict(kernel_size=1, stride=1, padding=0, output_padding=0, dilation=1, transposed=False)
        full.update(kw)
        geom = np.array(sum((list(porc._pair(full[key])) for key in ("kernel_size", "stride", "padding", "dilation",
                                                                     "output_padding")), []) + [int(full["transposed"])],
                        np.int32)
        save("g9_nd2col_" + name, x=xin, cols=cols, geom=geom)
        mine = porc.nd2col(xin, **kw)
        assert mine.shape == cols.shape and np.array_equal(mine, cols), name
        manifest["oracle_vs_reference"]["nd2col_" + name] = "bit-identical"
    # error behaviour (pac.py:77-78)
    try:
        ref_pac.conv2d(torch.zeros(1, 3, 4, 4), torch.zeros(1, 2, 3, 3, 4, 4), 3, 1, 1, 1)
        manifest["incompatible_kernel_ch"] = "no error"
    except ValueError as e:
        manifest["incompatible_kernel_ch"] = "ValueError: %s" % e
    manifest["torch"] = torch.__version__
    manifest["numpy"] = np.__version__
    with open(os.path.join(HERE, "golden_pac_conv_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("total fixture bytes", sum(v["bytes"] for v in manifest["files"].values()))


if __name__ == "__main__":
    main()
