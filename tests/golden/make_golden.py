#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference, only the .npz files written here.  Fixtures are data: inputs + the outputs
the reference produced for them.  Re-run:

    python tests/golden/make_golden.py

Also cross-checks, while the reference is importable, that
  * oracle/cspn_oracle.py (numpy restatement) and
  * oracle/ref_plumbing_torch.py (cpu_baseline op-mix port)
agree with the import, and records the observed deviations in golden_manifest.json.
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

from network.libs.post_process import CSPN_new, CSPN_ours  # noqa: E402  (reference)
from network.libs.base import pac as ref_pac                # noqa: E402  (reference)
from libs import metrics as ref_metrics                     # noqa: E402  (reference)
from oracle import cspn_oracle as orc                       # noqa: E402
from oracle import ref_plumbing_torch as plumb              # noqa: E402

torch.set_num_threads(8)
manifest = {}


def t(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x))


def ref_cspn3(g, d, s, T):
    with torch.no_grad():
        return CSPN_new.AffinityPropagate(T, 3)(t(g), t(d), t(s)).numpy()


def ref_pac_fwd(x, gd, s, T):
    with torch.no_grad():
        return CSPN_ours.AffinityPropagate(T)(t(x), t(gd), sparse_depth=t(s)).numpy()


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    same_nan = np.array_equal(np.isnan(a), np.isnan(b))
    if not fin.any():
        return 0.0 if same_nan else float("inf")
    e = float((np.abs(a - b)[fin] / np.maximum(np.abs(b[fin]), 1e-6)).max())
    return e if same_nan else float("inf")


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: v for k, v in arrs.items() if v is not None})
    manifest.setdefault("files", {})[name] = {
        "bytes": os.path.getsize(path),
        "arrays": {k: list(np.shape(v)) for k, v in arrs.items() if v is not None},
    }


def check(tag, got, want):
    e = relerr(got, want)
    manifest.setdefault("oracle_vs_reference_max_rel", {})[tag] = e
    assert e < 2e-5, (tag, e)


# ---------------------------------------------------------------- G1: small exhaustive
def g1():
    B, H, W = 2, 13, 17
    for C in (8, 12):
        g, d, s = orc.synthetic_inputs(seed=10 + C, B=B, H=H, W=W, C=C, sparse_samples=40)
        for T in (1, 2, 24):
            for sp in (None, s):
                tag = f"g1_c{C}_t{T}_{'sp' if sp is not None else 'nosp'}"
                out = ref_cspn3(g, d, sp, T)
                save(tag, guidance=g, blur=d, sparse=sp, out=out, T=np.int32(T))
                check(tag, orc.cspn3_forward(g, d, sp, T), out)
                with torch.no_grad():
                    p = plumb.cspn3_plumbing(t(g), t(d), t(sp), T).numpy()
                manifest.setdefault("plumbing_bit_identical", {})[tag] = bool(
                    np.array_equal(p, out, equal_nan=True))


# ---------------------------------------------------------------- G2: degenerate shapes
def g2():
    cases = {"1x1": (1, 1, 1), "1xW": (1, 1, 9), "Hx1": (1, 7, 1), "2x3": (2, 2, 3),
             "w65": (1, 5, 65), "w4": (1, 3, 4), "h70w68": (1, 70, 68)}
    for name, (B, H, W) in cases.items():
        g, d, s = orc.synthetic_inputs(seed=20, B=B, H=H, W=W, C=8, sparse_samples=max(1, H * W // 8))
        for sp in (None, s):
            tag = f"g2_{name}_{'sp' if sp is not None else 'nosp'}"
            out = ref_cspn3(g, d, sp, 5)
            save(tag, guidance=g, blur=d, sparse=sp, out=out, T=np.int32(5))
            check(tag, orc.cspn3_forward(g, d, sp, 5), out)
    # a pixel whose 8 neighbour gates are all exactly zero -> 0/0 = NaN that then spreads
    g, d, _ = orc.synthetic_inputs(seed=21, B=1, H=9, W=12, C=8)
    for k, (dy, dx) in enumerate(orc.CSPN3_OFFSETS):
        g[0, k, 4 + dy, 5 + dx] = 0.0
    out = ref_cspn3(g, d, None, 3)
    assert np.isnan(out).any()
    save("g2_zero_gates_nan", guidance=g, blur=d, out=out, T=np.int32(3))
    check("g2_zero_gates_nan", orc.cspn3_forward(g, d, None, 3), out)
    # negative sparse depth: mask = -1 quirk (CSPN_new.py:77-78, :90)
    g, d, s = orc.synthetic_inputs(seed=22, B=1, H=8, W=12, C=8, sparse_samples=30)
    s = s.copy()
    s[0, 0, 2, 3] = -1.5
    s[0, 0, 5, 8] = -0.25
    out = ref_cspn3(g, d, s, 6)
    save("g2_negative_sparse", guidance=g, blur=d, sparse=s, out=out, T=np.int32(6))
    check("g2_negative_sparse", orc.cspn3_forward(g, d, s, 6), out)


# ---------------------------------------------------------------- G3/G4: full frames from the hash generator
def g3_g4():
    for name, (H, W, sub) in {"g3_nyu": (228, 304, 1), "g4_kitti": (352, 1216, 8)}.items():
        for sparse in (False, True):
            g, d, s = orc.synthetic_inputs(seed=0, B=1, H=H, W=W, C=8,
                                           sparse_samples=500 if sparse else None)
            out = ref_cspn3(g, d, s, 24)
            tag = f"{name}_{'sp' if sparse else 'nosp'}"
            o64 = out.astype(np.float64)
            save(tag, out_sub=out[:, :, ::sub, ::sub], sub=np.int32(sub),
                 shape=np.array([1, H, W], np.int32), seed=np.int32(0),
                 sparse_samples=np.int32(500 if sparse else -1),
                 moments=np.array([o64.sum(), (o64 ** 2).sum()]), T=np.int32(24))
            check(tag, orc.cspn3_forward(g, d, s, 24), out)


# ---------------------------------------------------------------- G5: gradients of CSPN_new
def g5():
    B, H, W, T = 2, 9, 11, 5
    for C in (8, 12):
        g, d, s = orc.synthetic_inputs(seed=50 + C, B=B, H=H, W=W, C=C, sparse_samples=25)
        cot = orc.hash_normal(51, 9, (B, 1, H, W))
        for sp in (None, s):
            for dt in (torch.float64, torch.float32):
                torch.set_default_dtype(dt)   # the reference builds its ones-kernel in the default dtype
                gt = t(g).to(dt).requires_grad_(True)
                dtt = t(d).to(dt).requires_grad_(True)
                out = CSPN_new.AffinityPropagate(T, 3)(gt, dtt, None if sp is None else t(sp).to(dt))
                out.backward(t(cot).to(dt))
                torch.set_default_dtype(torch.float32)
                nm = "f64" if dt == torch.float64 else "f32"
                tag = f"g5_c{C}_{'sp' if sp is not None else 'nosp'}_{nm}"
                gg, gd = gt.grad.numpy(), dtt.grad.numpy()
                save(tag, guidance=g, blur=d, sparse=sp, cot=cot, grad_guidance=gg, grad_blur=gd,
                     T=np.int32(T))
                if dt == torch.float64:
                    og, od = orc.cspn3_backward(g, d, sp, cot, T, np.float64)
                    manifest.setdefault("oracle_grad_vs_reference_max_abs", {})[tag] = [
                        float(np.abs(og - gg).max()), float(np.abs(od - gd).max())]
                    assert np.abs(og - gg).max() < 1e-11 and np.abs(od - gd).max() < 1e-11


# ---------------------------------------------------------------- G6: PAC variant
def g6():
    B, H, W = 2, 20, 24
    for K, T in ((3, 24), (5, 12), (7, 4)):
        C = K * K - 1
        gd = orc.hash_normal(60 + K, 1, (B, C, H, W))
        x = orc.hash_uniform(60 + K, 2, (B, 1, H, W), 0.0, 10.0)
        s = orc.hash_sparse(60 + K, 3, x, 0.08)
        cot = orc.hash_normal(60 + K, 4, (B, 1, H, W))
        for sp in (None, s):
            tag = f"g6_k{K}_t{T}_{'sp' if sp is not None else 'nosp'}"
            out = ref_pac_fwd(x, gd, sp, T)
            # gradient oracle: the reference's own native_impl=True branch (pac.py:130-140) under autograd
            xt = t(x).double().requires_grad_(True)
            gt = t(gd).double().requires_grad_(True)
            sm = torch.softmax(gt, dim=1)
            kern = torch.zeros(B, C + 1, H, W, dtype=torch.float64)
            kern[:, :C // 2] = sm[:, :C // 2]
            kern[:, C // 2 + 1:] = sm[:, C // 2:]
            kern = kern.reshape(B, 1, K, K, H, W)
            cur = xt
            for _ in range(T):
                cur = ref_pac.conv2d(cur, kern, kernel_size=K, stride=1, padding=K // 2, dilation=1,
                                     native_impl=True)
                if sp is not None:
                    m = t(sp).double().sign()
                    cur = m * xt + (1 - m) * cur
            cur.backward(t(cot).double())
            assert relerr(cur.detach().numpy(), out) < 1e-5
            save(tag, x=x, guided=gd, sparse=sp, out=out, cot=cot, grad_x=xt.grad.numpy(),
                 grad_guided=gt.grad.numpy(), T=np.int32(T), K=np.int32(K))
            check(tag, orc.pac_forward(x, gd, sp, T), out)
            ox, og = orc.pac_backward(x, gd, sp, cot, T, np.float64)
            manifest.setdefault("oracle_grad_vs_reference_max_abs", {})[tag] = [
                float(np.abs(ox - xt.grad.numpy()).max()), float(np.abs(og - gt.grad.numpy()).max())]
            assert np.abs(ox - xt.grad.numpy()).max() < 1e-10
            assert np.abs(og - gt.grad.numpy()).max() < 1e-10
            with torch.no_grad():
                p = plumb.pac_plumbing(t(x), t(gd), t(sp), T).numpy()
            manifest.setdefault("plumbing_bit_identical", {})[tag] = bool(np.array_equal(p, out))
    # fp16 inputs through the reference (softmax in fp16, accumulation promoted to fp32)
    K, T = 5, 12
    gd = orc.hash_normal(66, 1, (1, 24, 16, 20)).astype(np.float16)
    x = orc.hash_uniform(66, 2, (1, 1, 16, 20), 0.0, 10.0).astype(np.float16)
    out = ref_pac_fwd(x, gd, None, T)
    save("g6_k5_t12_fp16", x=x, guided=gd, out=out, T=np.int32(T), K=np.int32(K))
    manifest["g6_fp16_out_dtype"] = str(out.dtype)


# ---------------------------------------------------------------- G7: metrics
def g7():
    H, W = 31, 37
    target = orc.hash_uniform(70, 1, (2, 1, H, W), 0.5, 10.0)
    noise = orc.hash_normal(70, 2, (2, 1, H, W)) * np.float32(0.1)
    pred = np.maximum(target + noise, np.float32(0.05)).astype(np.float32)
    inval = orc.hash_u24(70, 3, target.size).reshape(target.shape) < int(0.05 * 2 ** 24)
    target = np.where(inval, np.float32(0), target).astype(np.float32)
    r = ref_metrics.Result()
    r.evaluate(t(pred), t(target))
    vals = np.array([r.irmse, r.imae, r.mse, r.rmse, r.mae, r.absrel, r.lg10,
                     r.delta1, r.delta2, r.delta3], np.float64)
    save("g7_metrics", pred=pred, target=target, metrics=vals)
    got, n = orc.evaluate_metrics(pred, target)
    manifest["g7_oracle_rel"] = float(np.max(np.abs(got - vals) / np.abs(vals)))
    assert manifest["g7_oracle_rel"] < 1e-5 and n == int((target > 0).sum())


# ---------------------------------------------------------------- G8: hooked UNet tuple
def g8():
    try:
        from network import unet_cspn_nyu
    except Exception as e:  # pragma: no cover
        manifest["g8"] = f"skipped: {e!r}"
        return
    torch.manual_seed(0)
    net = unet_cspn_nyu.resnet50(pretrained=False).eval()
    cap = {}
    net.post_process_layer.register_forward_hook(lambda m, i, o: cap.update(i=i, o=o))
    rgb = orc.hash_uniform(80, 1, (1, 3, 228, 304), 0.0, 1.0)
    dep = orc.hash_uniform(80, 2, (1, 1, 228, 304), 0.5, 10.0)
    sp = orc.hash_sparse(80, 3, dep, 500.0 / (228 * 304))
    with torch.no_grad():
        net(torch.from_numpy(np.concatenate([rgb, sp], 1)))
    gdn, coarse, sparse = (x.numpy() for x in cap["i"])
    manifest["g8_head_stats"] = {"guidance_mean": float(gdn.mean()), "guidance_std": float(gdn.std()),
                                 "coarse_mean": float(coarse.mean()), "coarse_std": float(coarse.std()),
                                 "guidance_channels": int(gdn.shape[1])}
    ys, xs = slice(60, 156), slice(96, 224)      # 96 x 128 crop keeps the fixture small
    g16 = gdn[:, :, ys, xs].astype(np.float16)
    c16 = coarse[:, :, ys, xs].astype(np.float16)
    s16 = sparse[:, :, ys, xs].astype(np.float16)
    out = ref_cspn3(g16.astype(np.float32), c16.astype(np.float32), s16.astype(np.float32), 24)
    save("g8_unet_hook", guidance_f16=g16, blur_f16=c16, sparse_f16=s16, out=out, T=np.int32(24))
    check("g8_unet_hook", orc.cspn3_forward(g16.astype(np.float32), c16.astype(np.float32),
                                            s16.astype(np.float32), 24), out)


if __name__ == "__main__":
    for fn in (g1, g2, g3_g4, g5, g6, g7, g8):
        fn()
        print("done", fn.__name__, flush=True)
    manifest["torch"] = torch.__version__
    manifest["numpy"] = np.__version__
    with open(os.path.join(HERE, "golden_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    tot = sum(v["bytes"] for v in manifest["files"].values())
    print("total fixture bytes", tot)
