#!/usr/bin/env python3
"""Developer probe: which relative placements of the training path's big planes are slow?  The history planes (d history, G
history), the guidance gradient and S are carved out of ONE arena at controlled offsets (torch.empty is wrapped for the shapes the
path allocates); each of the three launches of a pass is timed with HIP events, 20 passes per placement.
    python tools/probes/train_layout_sweep.py [which=ghist|hist|gg] [step_kb=256] [n=33]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
dev = torch.device("cuda", 0)
B, H, W, T = 24, 228, 304, 24
which = sys.argv[1] if len(sys.argv) > 1 else "ghist"
step = int(sys.argv[2]) * 1024 if len(sys.argv) > 2 else 256 * 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 33
arena = torch.empty(1400 << 20, dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - arena.data_ptr()      # 2 MB aligned start
plane = B * H * W * 4
sizes = {"hist": T * plane, "ghist": T * plane, "gg": 12 * plane, "S": plane, "gd0": plane}
offsets = {}
real_empty = torch.empty
def carve(name, shape):
    o = offsets[name]
    return arena[o:o + sizes[name]].view(torch.float32).view(shape)
seen = {}
def fake_empty(*shape, **kw):
    shp = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
    if kw.get("dtype", torch.float32) == torch.float32 and kw.get("device") is not None:
        if shp == (T, B, H, W):
            k = "hist" if not seen.get("hist") else "ghist"
            seen["hist"] = True
            return carve(k, shp)
        if shp == (B, H, W):
            k = "S" if not seen.get("S") else "gd0"
            seen["S"] = True
            return carve(k, shp)
    return real_empty(*shape, **kw)
g = torch.randn(B, 12, H, W, device=dev, requires_grad=True)
d = (torch.rand(B, 1, H, W, device=dev) * 10).requires_grad_(True)
cot = torch.randn(B, 1, H, W, device=dev)
m = pkg.CSPN_new.AffinityPropagate(T, 3)
real_empty_like = torch.empty_like
def fake_empty_like(t, **kw):
    if t is g or (t.shape == g.shape and t.dtype == torch.float32):
        return carve("gg", tuple(g.shape))
    return real_empty_like(t, **kw)
def one_pass(ev):
    seen.clear()
    g.grad = None; d.grad = None
    ev[0].record(); out = m(g, d, None); ev[1].record()
    out.backward(cot); ev[2].record()
def place(delta):
    o = base
    lay = {}
    for name in ("hist", "S", "ghist", "gg", "gd0"):
        if name == which:
            o += delta
        lay[name] = o
        o += (sizes[name] + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    return lay
F.torch.empty = fake_empty; F.torch.empty_like = fake_empty_like
import time
for i in range(n):
    offsets.clear(); offsets.update(place(i * step))
    for _ in range(5): one_pass([torch.cuda.Event(enable_timing=True) for _ in range(3)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for _ in range(20):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        one_pass(ev); evs.append(ev)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e6
    fwd = sorted(e[0].elapsed_time(e[1]) * 1e3 for e in evs)[10]
    bwd = sorted(e[1].elapsed_time(e[2]) * 1e3 for e in evs)[10]
    print("%s +%6d KB: wall %.1f us  forward %.1f  backward (sweep + tail) %.1f" % (which, i * step >> 10, wall, fwd, bwd), flush=True)
