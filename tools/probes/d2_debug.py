"""Developer probe: which taps does the dot-product form apply?  Delta images through one step with uniform / ramp weights."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
dev = "cuda:0"
torch.manual_seed(0)
B, H, W, K = 1, 16, 32, 5
for T, S in ((1, 1), (2, 2), (4, 4), (12, 4)):
    g = torch.randn(B, 24, H, W, device=dev).half()
    x = (torch.rand(B, H, W, device=dev) * 10).half()
    with torch.no_grad():
        a = F.pac_forward_resident(g, x, None, T, steps_per_phase=S, step_form=F.STEP_DOT2)
        b = F.pac_forward_resident(g, x, None, T, steps_per_phase=S, step_form=F.STEP_FMA)
    F.ensure_resident_ok()
    d = (a.float() - b.float()).abs()
    print("T=%d S=%d max diff %.4f  at %s" % (T, S, float(d.max()), np.unravel_index(int(d.argmax()), d.shape)))
    if T == 1:
        print("diff map rows 0..5, cols 0..15:\n", np.array2string(d[0, :6, :16].cpu().numpy(), precision=3, suppress_small=True, max_line_width=200))
# delta probes, uniform weights
g = torch.zeros(B, 24, H, W, device=dev).half()
for (py, px) in ((8, 12), (8, 13)):
    x = torch.zeros(B, H, W, device=dev).half(); x[0, py, px] = 24.0
    with torch.no_grad():
        a = F.pac_forward_resident(g, x, None, 1, steps_per_phase=1, step_form=F.STEP_DOT2)
        b = F.pac_forward_resident(g, x, None, 1, steps_per_phase=1, step_form=F.STEP_FMA)
    F.ensure_resident_ok()
    print("delta at", (py, px), "dot2:\n", a[0, py - 3:py + 4, px - 4:px + 5].float().cpu().numpy())
    print("fma:\n", b[0, py - 3:py + 4, px - 4:px + 5].float().cpu().numpy())
# ramp weights: channel c has logit c*0.3 -> distinguishes the taps
g = (torch.arange(24, device=dev).float() * 0.3).view(1, 24, 1, 1).expand(B, 24, H, W).contiguous().half()
x = torch.zeros(B, H, W, device=dev).half(); x[0, 8, 12] = 100.0; x[0, 8, 21] = 100.0
with torch.no_grad():
    a = F.pac_forward_resident(g, x, None, 1, steps_per_phase=1, step_form=F.STEP_DOT2)
    b = F.pac_forward_resident(g, x, None, 1, steps_per_phase=1, step_form=F.STEP_FMA)
np.set_printoptions(precision=2, suppress=True, linewidth=220)
print("ramp dot2:\n", a[0, 5:12, 8:26].float().cpu().numpy())
print("ramp fma:\n", b[0, 5:12, 8:26].float().cpu().numpy())
