"""Developer probe (round 5): does the resident launch read LDS it never wrote?  Poison the LDS of every CU (tests/support
lds_poison) in front of the resident call and compare with the multi-launch schedule; report WHERE the bits differ."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
from oracle import c_oracle
from conftest import occupy_lib
c_oracle.build()
occ = occupy_lib()
occ.lds_poison.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
occ.lds_poison.restype = ctypes.c_int
DEV = "cuda:0"
sink = torch.zeros(4, dtype=torch.int32, device=DEV)
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def poison(pattern):
    assert occ.lds_poison(1024, pattern, 2000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


cases = [(8, 352, 1216, 24), (1, 352, 1216, 24), (24, 228, 304, 24), (3, 228, 304, 24), (30, 120, 160, 17), (5, 60, 64, 7)]
out_dir = os.path.join(ROOT, "gpurun_out", "r05_poison"); os.makedirs(out_dir, exist_ok=True)
report = []
m24 = {}
for (B, H, W, T) in cases:
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    for sparse in (False, True):
        g, d, s = c_oracle.synthetic_inputs(70 + B + T, B, H, W, 12, max(2, H * W // 140) if sparse else None)
        gt, dt, st = dev(g), dev(d), dev(s)
        with torch.no_grad():
            F.set_resident("off"); ref = m(gt, dt, st)
        for name, pat in (("none", None), ("nan", 0x7fc00000), ("one", 0x3f800000), ("big", 0x7149f2ca)):
            with torch.no_grad():
                F.set_resident("on")
                if pat is not None:
                    poison(pat)
                out = m(gt, dt, st)
            torch.cuda.synchronize(); F.check_resident_errors()
            diff = ~((out == ref) | (torch.isnan(out) & torch.isnan(ref)))
            idx = diff.nonzero().cpu().numpy()
            rp = F.resident_plan(B, H, W, T, int(sparse))
            rec = dict(case=[B, H, W, T], sparse=sparse, poison=name, n_bad=int(idx.shape[0]), plan={k: rp[k] for k in ("tiles_x", "tiles_y", "tile_w", "tile_h", "quads_per_thread", "steps_per_phase")})
            if idx.shape[0]:
                rec["rows"] = sorted(set(int(v) for v in idx[:, 2]))[:20]
                rec["imgs"] = sorted(set(int(v) for v in idx[:, 0]))
                rec["cols_minmax"] = [int(idx[:, 3].min()), int(idx[:, 3].max())]
            report.append(rec)
            print(json.dumps(rec), flush=True)
json.dump(report, open(os.path.join(out_dir, "poison_probe.json"), "w"), indent=1)
