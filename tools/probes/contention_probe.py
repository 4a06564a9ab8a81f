"""Developer probe: does the co-tenant kernel of tests/support/occupy.hip keep resident workgroups from becoming co-resident?"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import cspn_monodepth_amd as pkg
from cspn_monodepth_amd import functional as F
from conftest import occupy_lib
dev = "cuda:0"
occ = occupy_lib()
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
g = torch.randn(24, 12, 228, 304, device=dev)
d = torch.rand(24, 228, 304, device=dev) * 10
with torch.no_grad():
    for _ in range(3):
        F.forward_resident(g, d, None, 24, 0)
    torch.cuda.synchronize()
    for n_wg, lds, spin in ((40, 120 * 1024, 600), (64, 120 * 1024, 600), (40, 120 * 1024, 50), (128, 64 * 1024, 600), (40, 120 * 1024, 0)):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            e0.record(side)
            ok = occ.occupy(n_wg, lds, 500000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))
            e1.record(side)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        out = F.forward_resident(g, d, None, 24, 0, spin_limit=spin)
        eb.record()
        torch.cuda.synchronize()
        st = F._RES[0]
        print("tenant %3d wg x %3d KB: occupy ok=%d  tenant %.2f ms, resident launch %.3f ms, spin_limit %d -> host_err %s, NaN tiles: %s" % (
            n_wg, lds // 1024, ok, e0.elapsed_time(e1), ea.elapsed_time(eb), spin, list(st["host_err_np"][:2]), bool(torch.isnan(out).any())))
        try:
            F.ensure_resident_ok()
        except RuntimeError as e:
            print("   raised:", str(e)[:80])
        print("   fallbacks so far:", F.resident_fallbacks(), "mode", F._RESIDENT_MODE)
        F.set_resident("auto")

# ---- the test's loop: 20 scored forwards under a running tenant, spin limit 10
from cspn_monodepth_amd import evaluation as ev
m = pkg.CSPN_new.AffinityPropagate(24, 3)
tg = (d + 0.1).abs()
dd = d.unsqueeze(1).contiguous()
tt = tg.unsqueeze(1).contiguous()
for spin in (10, 3, 1):
    F._RESIDENT_SPIN_LIMIT = spin
    acc = ev.new_accumulator(dev)
    t0 = time.perf_counter()
    with torch.no_grad():
        for k in range(20):
            if k % 5 == 0:
                occ.occupy(64, 120 * 1024, 500000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))
            out = m.forward_scored(g, dd, None, tt, acc)
    t1 = time.perf_counter()
    torch.cuda.current_stream().synchronize()
    t2 = time.perf_counter()
    st = F._RES[0]
    print("spin %d: enqueue %.2f ms, drained after %.2f ms, host_err %s, journal %d, fallbacks %d, mode %s" % (
        spin, (t1 - t0) * 1e3, (t2 - t0) * 1e3, list(st["host_err_np"][:2]), len(st["journal"]), F.resident_fallbacks(), F._RESIDENT_MODE))
    total, _ = ev.all_gather_metric_sums(acc)
    print("   after gather: fallbacks %d, count %d" % (F.resident_fallbacks(), ev.finalize_metrics(total)["count"]))
    torch.cuda.synchronize()
    F.set_resident("auto")
F._RESIDENT_SPIN_LIMIT = 0
