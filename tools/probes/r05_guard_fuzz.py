"""Round-5 fuzz of the device-side guard (csrc/cspn_repair.hip): seeded random shapes, every resident launch forced to give up at its
first neighbour wait (one poll), the guard's re-computation compared with the multi-launch schedule — 3x3 inference (bit for bit),
scored inference under set_resident_guard("all") (depth bit for bit, sums to 1e-6), the training step (output and both gradients bit
for bit), K x K inference (FMA form: bit for bit).  usage: r05_guard_fuzz.py [cases]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F, evaluation as ev
from oracle import c_oracle
c_oracle.build()
warnings.simplefilter("ignore", RuntimeWarning)
DEV = "cuda:0"
dev = lambda a, dt=None: None if a is None else (torch.from_numpy(np.ascontiguousarray(a)).to(DEV) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dt))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(20250930)
bad = done = timed_out = 0


def forced(fn):
    prev = F._RESIDENT_SPIN_LIMIT
    F._RESIDENT_SPIN_LIMIT = 1
    try:
        return fn()
    finally:
        F._RESIDENT_SPIN_LIMIT = prev


for case in range(n_cases):
    B = int(rng.integers(1, 9)); H = int(rng.integers(6, 180)); W = int(4 * rng.integers(2, 80)); T = int(rng.integers(1, 31))
    if rng.random() < 0.15:
        W += int(rng.integers(1, 4))                       # odd widths: the row-padding path (W_valid)
    sparse = bool(rng.random() < 0.5); C = 12 if rng.random() < 0.7 else 8
    g, d, s = c_oracle.synthetic_inputs(3000 + case, B, H, W, C, max(2, H * W // 100) if sparse else None)
    gt, dt, st = dev(g), dev(d), dev(s)
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    n0 = F.resident_fallbacks()
    with torch.no_grad():
        F.set_resident("off"); ref = m(gt, dt, st)
        F.set_resident("on"); F.set_resident_guard(True)
        out = forced(lambda: m(gt, dt, st))
    ok = torch.equal(out, ref) or (torch.isnan(ref).any() and torch.equal(torch.nan_to_num(out, 7.0), torch.nan_to_num(ref, 7.0)))
    kinds = ["inference"]
    if W % 4 == 0 and B * H * W < 400000:
        tg = (dt + 0.1).contiguous()
        with torch.no_grad():
            F.set_resident("off"); a0 = ev.new_accumulator(DEV); m.forward_scored(gt, dt, st, tg, a0)
            F.set_resident("on"); F.set_resident_guard("all"); a1 = ev.new_accumulator(DEV)
            o2 = forced(lambda: m.forward_scored(gt, dt, st, tg, a1))
        s0, s1 = a0.sum(0).cpu().numpy(), a1.sum(0).cpu().numpy()
        ok = ok and torch.equal(o2, ref) and bool(np.allclose(s1, s0, rtol=1e-6, atol=1e-9, equal_nan=True))
        kinds.append("scored")
        F.set_resident_guard(True)
        cot = dev(c_oracle.hash_normal(4000 + case, 9, (B, 1, H, W)))
        res = []
        for lim in (0, 1):
            g_, d_ = gt.clone().requires_grad_(True), dt.clone().requires_grad_(True)
            F._RESIDENT_SPIN_LIMIT = lim
            try:
                o = m(g_, d_, st); o.backward(cot)
            finally:
                F._RESIDENT_SPIN_LIMIT = 0
            res.append((o.detach(), g_.grad, d_.grad))
        eq = lambda a, b: torch.equal(torch.nan_to_num(a, 7.0), torch.nan_to_num(b, 7.0))
        ok = ok and all(eq(a, b) for a, b in zip(*res))
        kinds.append("training")
    F.ensure_resident_ok()
    timed_out += int(F.resident_fallbacks() > n0)
    done += 1
    if not ok:
        bad += 1
        print("MISMATCH case %d: B=%d H=%d W=%d T=%d sparse=%s C=%d kinds=%s" % (case, B, H, W, T, sparse, C, kinds), flush=True)
# K x K inference
kbad = kdone = 0
for case in range(max(20, n_cases // 3)):
    K = int(rng.choice([3, 5])); B = int(rng.integers(1, 7)); H = int(rng.integers(4, 100)); W = int(8 * rng.integers(1, 30)); T = int(rng.integers(1, 14))
    sparse = bool(rng.random() < 0.5); f32 = bool(rng.random() < 0.3) and K == 5
    gd = c_oracle.hash_normal(5000 + case, 1, (B, K * K - 1, H, W)); x = c_oracle.hash_uniform(5000 + case, 2, (B, 1, H, W), 0.0, 10.0)
    s = c_oracle.hash_sparse(5000 + case, 3, x, 0.02) if sparse else None
    tdt = torch.float32 if f32 else torch.float16
    xt, gt, st = dev(x, tdt), dev(gd, tdt), dev(s, tdt)
    rp = F.kres_plan(K, B, H, W, T, int(sparse), 0, 0, 0, F.CSPN_F32 if f32 else F.CSPN_F16)
    if rp is None:
        continue
    mm = pkg.CSPN_ours.AffinityPropagate(T, plan=dict(steps_per_launch=rp["steps_per_phase"]), state_dtype=None)
    with torch.no_grad():
        F.set_resident("off")
        try:
            ref = mm(xt, gt, sparse_depth=st)[:, 0]
        except RuntimeError:
            continue                                       # (no multi-launch tiling for this phase length: skip)
        F.set_resident("on")
        out = forced(lambda: F.pac_forward_resident(gt, xt[:, 0].contiguous(), None if st is None else st[:, 0].contiguous(), T, step_form=F.STEP_FMA))
    F.ensure_resident_ok()
    kdone += 1
    if not torch.equal(out, ref):
        kbad += 1
        print("K x K MISMATCH case %d: K=%d B=%d H=%d W=%d T=%d sparse=%s f32=%s: %d px" % (case, K, B, H, W, T, sparse, f32, int((out != ref).sum())), flush=True)
# K = 5 fp16 training forms: the reverse sweep (forced against un-forced, bit for bit) and the forward with history (against
# cspn_pac_prepare + one-step launches, bit for bit, whenever a tile really gave up)
tbad = tdone = 0
for case in range(max(20, n_cases // 3)):
    B = int(rng.integers(1, 30)); H = int(rng.integers(6, 120)); W = int(8 * rng.integers(1, 40)); T = int(rng.integers(2, 20))
    sparse = bool(rng.random() < 0.5)
    rp = F.kres_plan(5, B, H, W, T, int(sparse))
    if rp is None or rp["quads_per_thread"] != 1 or 2 * T > 54:
        continue
    gd = c_oracle.hash_normal(7000 + case, 1, (B, 24, H, W)); x = c_oracle.hash_uniform(7000 + case, 2, (B, 1, H, W), 0.0, 10.0)
    s = c_oracle.hash_sparse(7000 + case, 3, x, 0.02) if sparse else None
    gt, xt = dev(gd, torch.float16), dev(x, torch.float16)[:, 0].contiguous()
    st = None if s is None else dev(s, torch.float16)[:, 0].contiguous()
    cot = dev(c_oracle.hash_normal(7000 + case, 9, (B, H, W))).half()
    F.set_resident("on"); F.set_resident_guard(True)
    with torch.no_grad():
        wk0, _ = F.pac_prepare(gt)
        g0, want = F.pac_transposed_resident(wk0, cot, st, T)
        g1, got = forced(lambda: F.pac_transposed_resident(wk0, cot, st, T))
        ok = torch.equal(torch.nan_to_num(got, 7.0), torch.nan_to_num(want, 7.0)) and torch.equal(g0, g1)
        F.ensure_resident_ok()                            # (the sweep's time-out is counted where the host next looks: look now)
        n0 = F.resident_fallbacks()
        out1, hist1, wk1 = forced(lambda: F.pac_forward_resident_history(gt, xt, st, T))
        F.ensure_resident_ok()
        if F.resident_fallbacks() > n0:
            F.set_resident("off")
            try:
                _, hist0 = F.propagate(wk0, xt, st, 5, T, F.BLEND_SPARSE if sparse else F.BLEND_NONE, keep_history=True,
                                       plan=F.dtype_default_plan(5, wk0.dtype, dict(steps_per_launch=1)))
                ok = ok and torch.equal(wk1, wk0) and torch.equal(torch.nan_to_num(hist1.float(), 7.0), torch.nan_to_num(hist0.float(), 7.0))
            except RuntimeError:
                pass
            F.set_resident("on")
    tdone += 1
    if not ok:
        tbad += 1
        print("K = 5 TRAINING MISMATCH case %d: B=%d H=%d W=%d T=%d sparse=%s" % (case, B, H, W, T, sparse), flush=True)
print("guard fuzz, K = 5 fp16 training forms: %d cases, %d mismatches" % (tdone, tbad))
print("guard fuzz: %d 3x3 cases (%d with a forced time-out), %d mismatches; %d K x K cases, %d mismatches" % (done, timed_out, bad, kdone, kbad))
