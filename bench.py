#!/usr/bin/env python3
"""bench.py — the hot-path benchmark (BASELINE.json metric: depth-maps/sec at 304x228 b=24, 24 CSPN iters;
achieved HBM GB/s vs roofline).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM: the module
forward `AffinityPropagate(24, 3)(guidance[24,12,228,304], coarse[24,1,228,304])` (prepare + 24 propagation
steps, all in libcspn_hip.so) plus the fused depth-metrics reduction of the refined batch.  N > 1: one
process per GPU, every rank refines its own batch (weak scaling, no collective inside the loop), one RCCL
all-gather of the 10 metric sums at the end of the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cspn_monodepth_amd as pkg                      # noqa: E402
from cspn_monodepth_amd import functional as F        # noqa: E402

if int(os.environ.get("LOCAL_RANK", "0")) == 0:
    pkg._lib.build()                                  # no-op when libcspn_hip.so is current

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable

WORKLOADS = {
    # name: (B per GPU, H, W, K, T, dtype, guidance channels)
    "nyu": dict(B=24, H=228, W=304, K=3, T=24, dtype="f32", C=12,
                name="NYU-v2 304x228 batch=24, 24-iter 3x3 CSPN (BASELINE config 2)"),
    "kitti": dict(B=8, H=352, W=1216, K=3, T=24, dtype="f32", C=12,
                  name="KITTI 1216x352 batch=8 sharded over the ranks, 24-iter 3x3 CSPN (BASELINE config 4)"),
    "pac5f32": dict(B=24, H=228, W=304, K=5, T=12, dtype="f32", C=24,
                    name="NYU-v2 304x228 batch=24, 5x5 softmax affinity, 12 iters, fp32"),
    "pac5": dict(B=24, H=228, W=304, K=5, T=12, dtype="f16", C=24,
                 name="NYU-v2 304x228 batch=24, 5x5 softmax affinity, 12 iters, fp16 (BASELINE config 3)"),
    "train": dict(B=3, H=228, W=304, K=3, T=24, dtype="f32", C=12,
                  name="NYU-v2 training step: ResNet-50 UNet + affinity head (stock PyTorch-ROCm) + HIP CSPN forward/backward, "
                       "batch 3 per GPU (BASELINE config 5: global batch 24 on 8 GPUs, DDP + SyncBatchNorm)"),
}


def parse_plan(txt):
    if not txt:
        return None
    if txt == "auto":
        return "auto"
    keys = ("steps_per_launch", "tile_w", "tile_h", "quads_per_thread", "threads")
    return {k: int(v) for k, v in zip(keys, txt.split(","))}


def make_inputs(wl, B, device, seed, sparse):
    """SURVEY.md §8(d) value distributions (synthetic; no dataset on the box)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    dt = torch.float16 if wl["dtype"] == "f16" else torch.float32
    g = torch.randn(B, wl["C"], wl["H"], wl["W"], device=device, generator=gen).to(dt)
    d = (torch.rand(B, 1, wl["H"], wl["W"], device=device, generator=gen) * 10).to(dt)
    s = None
    if sparse:
        keep = torch.rand(B, 1, wl["H"], wl["W"], device=device, generator=gen) < 500.0 / (wl["H"] * wl["W"])
        s = (d * keep).to(dt)
    noise = torch.randn(B, 1, wl["H"], wl["W"], device=device, generator=gen) * 0.1
    target = (d.float() + noise).clamp_min(0.05)
    target = torch.where(torch.rand(B, 1, wl["H"], wl["W"], device=device, generator=gen) < 0.05,
                         torch.zeros_like(target), target).to(dt)
    return g, d, s, target


def load_pmc_traffic(tag, schedule="fused", launches_of_rarest=1):
    """HBM traffic per launch from the committed rocprofv3 --pmc summary of this workload (profiles/*_pmc_traffic_<tag>
    .json, newest round first).  `schedule`: "fused" = the passes of the default schedule (the weight-resident launch
    where it applies), "multi" = the passes with CSPN_RESIDENT=off.  `stale` = the kernel sources changed since."""
    import glob
    out = {"source": None, "stale": None, "step_bytes_per_launch": None, "fused_bytes_per_forward": None,
           "fused_per_launch": None, "per_kernel_per_forward": None, "sq": None}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % tag)), reverse=True)
    if not files:
        return out
    try:
        j = json.load(open(files[0]))
    except Exception:
        return out
    out["source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, gfx950-corrected%s)" % (
        os.path.basename(files[0]), "; measured at commit %s" % j["commit"] if j.get("commit") else "")
    try:
        out["stale"] = (j.get("source_digest") != pkg._lib.code_digest()) if j.get("source_digest") else None
    except Exception:
        out["stale"] = None
    pk = j.get("per_kernel", {})
    sched = pk.get(schedule) or pk.get("fused", {})
    # every kernel of the forward counts: the prepare / softmax pass, each propagation launch, the metrics reduction.
    # A forward runs its prepare (or its metrics, or its single resident launch) exactly once, so the smallest dispatch
    # count over the schedule's kernels is the number of forwards profiled; `launches_of_rarest` corrects that when even
    # the rarest kernel runs several times per forward (a resident schedule chunked over several launches).
    allk = [(k, v) for k, v in sched.items() if k.startswith("cspn") and "hbm_bytes_corrected" in v]
    fused = [(k, v) for k, v in allk if k.startswith(("cspn_prop", "cspn3_resident", "cspnk_resident", "cspnk_d2"))]
    if fused:
        n_fwd = min(v["_dispatches_FETCH_SIZE"] for _, v in allk) / max(1, launches_of_rarest)
        out["fused_per_launch"] = {k: v["hbm_bytes_corrected"] for k, v in fused}
        out["per_kernel_per_forward"] = {k: [v["hbm_bytes_corrected"], v["_dispatches_FETCH_SIZE"] / n_fwd] for k, v in allk}
        out["fused_bytes_per_forward"] = sum(v["hbm_bytes_corrected"] * v["_dispatches_FETCH_SIZE"] for _, v in allk) / n_fwd
    step = [(k, v) for k, v in pk.get("step", {}).items() if k.startswith("cspn_prop") and "hbm_bytes_corrected" in v]
    if step:                                                             # the plain streaming instance dominates
        out["step_bytes_per_launch"] = max(step, key=lambda kv: kv[1]["_dispatches_FETCH_SIZE"])[1]["hbm_bytes_corrected"]
    sq = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_%s.json" % tag)), reverse=True)
    if sq:
        try:
            out["sq"] = dict(json.load(open(sq[0])), source="profiles/" + os.path.basename(sq[0]))
        except Exception:
            pass
    return out


def load_train_traffic(tag):
    """Measured HBM bytes of one forward+backward of the module (tools/run_train_leg.py under rocprofv3 --pmc), from the
    committed summary profiles/r*_pmc_traffic_<tag>bwd.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%sbwd.json" % tag)), reverse=True)
    if not files:
        return None
    try:
        j = json.load(open(files[0]))
        ks = {k: v for k, v in j["per_kernel"]["fused"].items() if "hbm_bytes_corrected" in v}
        iters = min(v["_dispatches_FETCH_SIZE"] for k, v in ks.items() if k.startswith("cspn_grad_tail"))
        per = {k: [v["hbm_bytes_corrected"], v["_dispatches_FETCH_SIZE"] / iters] for k, v in ks.items()}
        return {"bytes_per_step": sum(b * n for b, n in per.values()), "per_kernel": per,
                "source": "profiles/%s%s" % (os.path.basename(files[0]), "; measured at commit %s" % j["commit"] if j.get("commit") else ""),
                "stale": (j.get("source_digest") != pkg._lib.code_digest()) if j.get("source_digest") else None}
    except Exception:
        return None


_ORIG_AFFINITY = None        # the affinity mask this process was started with (bind_cpus narrows it for the GPU legs)


def bind_cpus(local_rank, local_world, mode):
    """Keep the host threads of a rank (Python, the autograd engine's worker, the HIP runtime's) on a few ADJACENT cores.
    On the 256-thread hosts of the GPU boxes the scheduler otherwise spreads them over the sockets' CCDs, and a launch-rate-bound
    leg then depends on where they happen to land: the training-shaped leg read 213-217 us per pass in some processes and 265-290 us
    in others with IDENTICAL kernel times (rocprofv3: 86 + 61 + 60 us in every run) — bound to four cores it reads 213-217 us in
    every run, and the headline value gains ~2 % (same-box A/B, five alternating runs; round 4).  Round 5 removed the cause (the leg
    is GPU-bound now, see --cpu-bind) and the binding is opt-in.
    What `numactl` / `taskset` in a launch script would do; --cpu-bind off leaves the mask alone.  (The block is the rank's
    share of the FIRST allowed cores on purpose: picking the idlest block of the moment — /proc/stat over 40 ms — was tried and
    is worse, twice it chose cores 4-7 of a box and the leg read 600-720 us: idle cores sit in deep C-states and the autograd
    engine's hand-offs pay their wake-up latency; cores 0-3 are kept awake by the system's own housekeeping.)"""
    global _ORIG_AFFINITY
    if mode == "off":
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    _ORIG_AFFINITY = set(allowed)
    k = 4 if len(allowed) >= 4 * max(local_world, 1) else len(allowed) // max(local_world, 1)
    if k < 2:
        return None
    mine = allowed[(local_rank % max(local_world, 1)) * k:(local_rank % max(local_world, 1) + 1) * k]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


class full_affinity(object):
    """The CPU baseline runs on every core the process was given, not on the GPU legs' four."""

    def __enter__(self):
        self.cur = None
        if _ORIG_AFFINITY is not None:
            try:
                self.cur = os.sched_getaffinity(0)
                os.sched_setaffinity(0, _ORIG_AFFINITY)
            except OSError:
                self.cur = None

    def __exit__(self, *exc):
        if self.cur is not None:
            try:
                os.sched_setaffinity(0, self.cur)
            except OSError:
                pass
        return False


def host_cpu_budget():
    """Cores this process may use: the scheduler affinity mask and the cgroup CPU quota (v2 cpu.max / v1 cfs quota), next
    to os.cpu_count() (which counts the machine's threads whatever the container is allowed)."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(_ORIG_AFFINITY) if _ORIG_AFFINITY is not None else len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = logical
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.999)))
    return {"os_cpu_count": logical, "sched_affinity": affinity, "cgroup_cpu_quota": quota, "usable": usable}


def cpu_baseline(wl, budget_s=16.0):
    """The reference's CPU op mix (oracle/ref_plumbing_torch.py, a port validated bit-identical to the imported
    reference) on this box's host cores, on a bounded sample of the same workload: single frames and a
    4-frame batch of the workload's shape, at up to four thread counts (1/8/16/32, capped by the core count) plus one
    all-threads leg in a child process with a timeout (oneDNN's 5-D conv path scales badly with both
    batch and threads); the best rate is reported."""
    from oracle import ref_plumbing_torch as plumb
    from oracle import c_oracle
    host = host_cpu_budget()
    cores = host["usable"]          # what this process may actually run on — not os.cpu_count()
    H, W, T = wl["H"], wl["W"], wl["T"]
    torch.manual_seed(0)

    def make(b):
        if wl["K"] == 3:
            g, d = torch.randn(b, 8, H, W), torch.rand(b, 1, H, W) * 10
            return lambda: plumb.cspn3_plumbing(g, d, None, T)
        g, d = torch.randn(b, wl["C"], H, W), torch.rand(b, 1, H, W) * 10
        return lambda: plumb.pac_plumbing(d, g, None, T)

    best = None
    tried = []
    t_begin = time.perf_counter()
    with torch.no_grad():
        for threads in sorted({1, min(cores, 8), min(cores, 16), min(cores, 32)}):
            torch.set_num_threads(threads)
            for b in (1, 4):
                if time.perf_counter() - t_begin > budget_s:
                    break
                fn = make(b)
                fn()                                                   # warm-up
                times = []
                t_leg = time.perf_counter()
                while len(times) < 5 and time.perf_counter() - t_leg < budget_s / 4:
                    t0 = time.perf_counter()
                    fn()
                    times.append(time.perf_counter() - t0)
                rate = b / sorted(times)[len(times) // 2]
                tried.append({"threads": threads, "batch": b, "maps_per_s": rate, "reps": len(times),
                              "maps_per_s_min": b / max(times), "maps_per_s_max": b / min(times)})
                if best is None or rate > best[0]:
                    best = (rate, threads, b, len(times), b / max(times), b / min(times))
    # all host threads (SURVEY.md 8d asks for os.cpu_count() next to 1): oneDNN's 5-D conv path can take minutes per
    # frame at 256 threads, so this leg runs in a child process that is killed after `all_core_timeout_s`
    all_cores = None
    if cores > 32:
        import subprocess
        code = ("import sys, time, json, torch; sys.path.insert(0, %r); from oracle import ref_plumbing_torch as p; "
                "torch.set_num_threads(%d); torch.manual_seed(0); "
                "g, d = torch.randn(1, %d, %d, %d), torch.rand(1, 1, %d, %d) * 10; "
                "f = (lambda: p.cspn3_plumbing(g, d, None, %d)) if %d == 3 else (lambda: p.pac_plumbing(d, g, None, %d)); "
                "f(); ts = []\n"
                "for _ in range(3):\n    t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)\n"
                "print(json.dumps({'maps_per_s': 1.0 / sorted(ts)[1], 'reps': 3}))" % (
                    ROOT, cores, 8 if wl["K"] == 3 else wl["C"], H, W, H, W, T, wl["K"], T))
        all_core_timeout_s = 45
        try:
            with torch.no_grad():
                cp = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=all_core_timeout_s)
            all_cores = dict(json.loads(cp.stdout.strip().splitlines()[-1]), threads=cores, batch=1)
            tried.append({"threads": cores, "batch": 1, "maps_per_s": all_cores["maps_per_s"], "reps": 3})
            if all_cores["maps_per_s"] > best[0]:
                best = (all_cores["maps_per_s"], cores, 1, 3, all_cores["maps_per_s"], all_cores["maps_per_s"])
        except subprocess.TimeoutExpired:
            all_cores = {"threads": cores, "batch": 1, "timed_out_after_s": all_core_timeout_s,
                         "maps_per_s_upper_bound": 4.0 / all_core_timeout_s}
        except Exception as e:                                         # noqa: BLE001
            all_cores = {"threads": cores, "error": repr(e)[:200]}
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    one = [t for t in tried if t["threads"] == 1]
    # `value` is the median of the best leg; the same box class read 162 and 192 maps/s in two runs of round 5, so the spread is on the
    # line too: min-max over the repetitions of the best leg, and over the medians of every leg tried
    out = {"value": best[0], "value_range": [best[4], best[5]],
           "range_over_legs": [min(t["maps_per_s"] for t in tried), max(t["maps_per_s"] for t in tried)],
           "unit": "depth-maps/s", "cores": best[1], "kind": "port", "host_cores": cores, "host_cpu_budget": host,
           "cpu_model": model, "single_thread_value": max(t["maps_per_s"] for t in one) if one else None,
           "all_threads_leg": all_cores,
           "sample": "%d forward(s) of %d frame(s) %dx%d, T=%d (median, after warm-up) with %d of %d host threads; "
                     "PyTorch %s CPU op-mix port of the reference (pad/cat/conv3d-ones/div); legs tried: %s" % (
                         best[3], best[2], W, H, T, best[1], cores, torch.__version__, json.dumps(tried))}
    try:                                                               # scalar C oracle on one core, for calibration
        if wl["K"] == 3:
            gn, dn, _ = c_oracle.synthetic_inputs(0, 4, H, W, 8, None)
            t0 = time.perf_counter()
            c_oracle.cspn3_forward(gn, dn, None, T)
            out["c_oracle_1core_maps_per_s"] = 4 / (time.perf_counter() - t0)
    except Exception as e:  # pragma: no cover
        out["c_oracle_error"] = repr(e)
    return out


def stock_ops_same_gpu(wl, g, d, s, device, hip_out=None, reps=5):
    """The reference runs this module on the GPU through stock ATen ops (libs/trainers/single_gpu_trainer.py:72-77: `model(input)`
    on cuda); its Python cannot travel, so — like the CPU leg — the baseline is the op-mix port (oracle/ref_plumbing_torch.py:
    pad / cat / conv3d-with-ones / div per step, validated bit-identical to the imported reference on CPU) run on THIS GPU at the
    workload's full batch, outside the timed region.  What a maintainer deciding on the import swap compares `value` with."""
    from oracle import ref_plumbing_torch as plumb
    H, W, T, K, B = wl["H"], wl["W"], wl["T"], wl["K"], g.shape[0]
    with torch.no_grad():
        gf, df = g.float(), d.float()
        sf = None if s is None else s.float()
        if K == 3:
            fn = lambda: plumb.cspn3_plumbing(gf, df, sf, T)          # noqa: E731  (reads channels 0..7 of the 12, as the reference)
        else:
            fn = lambda: plumb.pac_plumbing(df, gf, sf, T)            # noqa: E731
        ref = fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        out = {"maps_per_s": B / sorted(times)[len(times) // 2], "maps_per_s_range": [B / max(times), B / min(times)],
               "ms_per_forward": sorted(times)[len(times) // 2] * 1e3, "batch": B, "reps": reps, "dtype": "f32",
               "kind": "port", "what": "oracle/ref_plumbing_torch.py (the reference's op mix: per step 8 pads + cat, a product, two "
                                       "conv3d-with-ones channel sums, a divide, a crop%s) on cuda:0 through stock PyTorch-ROCm ops, "
                                       "PyTorch %s; outside the timed region" % (", the sparse blend" if s is not None else "", torch.__version__)}
        # device kernels per forward (what the one resident launch replaces); ATen operator calls as the fallback figure
        try:
            from torch.utils._python_dispatch import TorchDispatchMode

            class _Count(TorchDispatchMode):
                n = 0

                def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                    _Count.n += 1
                    return func(*args, **(kwargs or {}))
            with _Count():
                fn()
            out["aten_ops_per_forward"] = _Count.n
        except Exception as e:                                         # noqa: BLE001
            out["aten_ops_per_forward"] = None
            out["aten_count_error"] = repr(e)[:120]
        try:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                fn()
                torch.cuda.synchronize()
            nk = sum(1 for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA"))
            out["kernel_launches_per_forward"] = nk if nk > 0 else None
        except Exception as e:                                         # noqa: BLE001
            out["kernel_launches_per_forward"] = None
            out["kernel_count_error"] = repr(e)[:120]
        if hip_out is not None:
            a, b = hip_out.float().reshape(-1), ref.reshape(-1)
            fin = torch.isfinite(a) & torch.isfinite(b)
            out["max_rel_diff_hip_vs_stock_ops"] = float(((a - b).abs()[fin] / b.abs().clamp_min(1e-6)[fin]).max())
    return out


def run_train(args, wl, world, rank, local_rank, device):
    """BASELINE config 5: one optimiser step of the re-hosted unet_cspn_nyu topology (cspn_monodepth_amd/network/
    unet_cspn_nyu.py: stock PyTorch-ROCm convolutions + the HIP un-pooling + the HIP CSPN module in forward AND
    backward), MaskedL1Loss (libs/criterion/criteria.py:27-39), SGD (main.py:72-74), synthetic RGB-D batch of
    `B` frames per GPU; N > 1: DistributedDataParallel over RCCL + nn.SyncBatchNorm (weak scaling: the global batch
    grows with N, config 5 = 3 x 8).  Reports depth-maps/s through the whole step and the CSPN module's share of it
    (HIP events recorded by module hooks on the launch stream, inside the timed region)."""
    from cspn_monodepth_amd.network import unet_cspn_nyu
    import torch.nn as nn
    B, H, W = wl["B"], wl["H"], wl["W"]
    # the reference's trainer sets cudnn.benchmark (main.py:37): MIOpen then picks every convolution algorithm by timing —
    # 36.5 -> 28.3 ms per step here, but ~8 minutes of searching on a fresh box (no tuning database travels), hence opt-in
    torch.backends.cudnn.benchmark = (args.conv_autotune == "on")
    # ... and what that search found on an MI355X ships with the package (MIOpen's own find-db / perf-db files): immediate
    # mode then picks the tuned solvers at once — 28.3 ms per step, no find phase in the first steps (--conv-db off: 36.6 ms)
    conv_db = None
    if args.conv_db == "shipped":
        from cspn_monodepth_amd.network import conv_tuning
        conv_db = conv_tuning.use_tuned_conv_db(rank=local_rank)
    torch.manual_seed(0)
    model = unet_cspn_nyu.resnet50(reference_state_dict=False, cspn_plan=parse_plan(args.plan), affinity_channels=args.affinity_channels).to(device)
    if args.memory_format == "channels_last":      # stock-op layout choice only: the HIP ops take their planes contiguous
        model = model.to(memory_format=torch.channels_last)
    if world > 1:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
        ddp = nn.parallel.DistributedDataParallel(model, device_ids=[device.index] if args.backend == "nccl" else None,
                                                  gradient_as_bucket_view=True)
    else:
        ddp = model
    # main.py:72-74's optimiser; `fused` = PyTorch's one-pass implementation of the same update (three foreach passes over 218 M
    # parameters otherwise: 2.3 of the step's 22.5 ms of GPU time)
    sgd_kw = {"fused": True} if args.sgd == "fused" else {}
    opt = torch.optim.SGD(model.parameters(), lr=0.001, momentum=0.9, weight_decay=1e-4, **sgd_kw)
    gen = torch.Generator(device=device).manual_seed(4321 + rank)
    depth = (torch.rand(B, 1, H, W, device=device, generator=gen) * 9.5 + 0.5)
    rgb = torch.rand(B, 3, H, W, device=device, generator=gen)
    sparse = depth * (torch.rand(B, 1, H, W, device=device, generator=gen) < 500.0 / (H * W))
    x = torch.cat([rgb, sparse], 1)
    if args.memory_format == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
    target = torch.where(torch.rand(B, 1, H, W, device=device, generator=gen) < 0.05, torch.zeros_like(depth), depth)

    cspn = model.post_process_layer
    ev = {"f0": [], "f1": [], "b0": [], "b1": []}
    timing = {"on": False}

    def rec(key):
        def hook(*_):
            if timing["on"]:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev[key].append(e)
        return hook
    cspn.register_forward_pre_hook(rec("f0"))
    cspn.register_forward_hook(rec("f1"))
    cspn.register_full_backward_pre_hook(rec("b0"))
    cspn.register_full_backward_hook(rec("b1"))

    losses = []

    def step():
        pred = ddp(x)
        valid = target > 0
        loss = ((target - pred).abs() * valid).sum() / valid.sum()          # MaskedL1Loss
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.detach())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    graphed = False
    if args.graph == "on" and world == 1:
        # the whole optimiser step (forward, loss, backward, SGD) as ONE HIP graph: ~790 kernels per step, most of them a few
        # microseconds long, leave the GPU idle a fifth of the time when they are launched one by one.  Static input batch,
        # gradients allocated inside the capture (set_to_none), the weight-resident CSPN launches captured with their control-word
        # memsets (functional._resident_launch).  Single process only: DDP's bucketed all-reduce is not captured here.
        eager_step = step
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                pred = ddp(x)
                valid = target > 0
                static_loss = ((target - pred).abs() * valid).sum() / valid.sum()
                static_loss.backward()
                opt.step()

            def step():                                                      # noqa: F811
                graph.replay()
                losses.append(static_loss.detach().clone())
            graphed = True
            for _ in range(3):
                step()
        except Exception as e:                                              # noqa: BLE001
            print("bench: whole-step graph capture failed (%r): eager steps" % (e,), file=sys.stderr)
            step = eager_step
            torch.cuda.synchronize()
    timing["on"] = True
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    timing["on"] = False
    if graphed:                      # replays bypass the module's own checks: look at the resident launches' error word now
        pkg.functional.mark_resident_pending()
        pkg.functional.ensure_resident_ok(device)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    fwd_us = sorted(a.elapsed_time(b) * 1e3 for a, b in zip(ev["f0"], ev["f1"]))
    bwd_us = sorted(a.elapsed_time(b) * 1e3 for a, b in zip(ev["b0"], ev["b1"]))
    infer = None
    if args.infer_batch > 0 and world == 1:
        # the eval pipeline's model(input) (libs/trainers/single_gpu_trainer.py:129): UNet + CSPN, eval mode, no autograd
        Bi = args.infer_batch
        xi = torch.cat([torch.rand(Bi, 3, H, W, device=device, generator=gen),
                        (torch.rand(Bi, 1, H, W, device=device, generator=gen) * 9.5 + 0.5) *
                        (torch.rand(Bi, 1, H, W, device=device, generator=gen) < 500.0 / (H * W))], 1)
        model.eval()
        with torch.no_grad():
            for _ in range(3):
                model(xi)
            torch.cuda.synchronize()
            ni = max(5, args.steps // 2)
            ti = time.perf_counter()
            for _ in range(ni):
                model(xi)
            torch.cuda.synchronize()
            infer = {"batch": Bi, "ms_per_step": (time.perf_counter() - ti) / ni * 1e3, "steps": ni}
            infer["maps_per_s"] = Bi / (infer["ms_per_step"] / 1e3)
        model.train()
    med = lambda v: v[len(v) // 2] if v else None                           # noqa: E731
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        lv = [float(v) for v in losses]
        res = {"metric": "depth-maps/sec through one training step (%s)" % wl["name"],
               "value": B * world * args.steps / elapsed, "unit": "depth-maps/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic RGB-D batch (random-init weights; no dataset / checkpoint on the box)",
               "config": {"workload": wl["name"], "batch_per_gpu": B, "global_batch": B * world, "H": H, "W": W,
                          "prop_time": wl["T"], "optimizer": "SGD(momentum 0.9, wd 1e-4%s)" % (", fused" if args.sgd == "fused" else ""), "loss": "MaskedL1",
                          "parameters": int(sum(p.numel() for p in model.parameters())),
                          "conv_autotune": args.conv_autotune, "memory_format": args.memory_format, "hip_graph": graphed,
                          "conv_db": ("shipped (cspn_monodepth_amd/network/miopen_db)" if conv_db else
                                      ("environment" if os.environ.get("MIOPEN_USER_DB_PATH") else "none")),
                          "parallelism": "DDP x%d over RCCL + SyncBatchNorm" % world if world > 1 else "single GPU"},
               "roofline": None,
               "cspn_module": {"forward_us_p50": med(fwd_us), "backward_us_p50": med(bwd_us),
                               "share_of_step": ((med(fwd_us) or 0) + (med(bwd_us) or 0)) / (ms * 1e3),
                               "note": "HIP events from module hooks around CSPN_new.AffinityPropagate's forward (derive launch "
                                       "+ history) and backward (reverse sweep + fused tail); the rest of the step is stock "
                                       "MIOpen/rocBLAS convolutions, batch-norm, the un-pooling kernel, SGD"},
               "loss_first_last": [lv[0], lv[-1]], "optimiser_steps_run": len(lv)}
        res["config"]["affinity_channels"] = args.affinity_channels
        if infer is not None:
            res["inference_step"] = infer
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="nyu", choices=sorted(WORKLOADS))
    ap.add_argument("--sparse", action="store_true", help="pass a 500-sample sparse depth (48 B/px/step)")
    ap.add_argument("--plan", default="", help="S,tile_w,tile_h,quads_per_thread,threads (default: built-in)")
    ap.add_argument("--conv-db", choices=("shipped", "off"), default="shipped",
                    help="--workload train: point MIOpen at the tuning database shipped with the package (the result of the "
                         "reference's cudnn.benchmark search on an MI355X) unless MIOPEN_USER_DB_PATH is already set")
    ap.add_argument("--sgd", choices=("foreach", "fused"), default="fused", help="--workload train: torch.optim.SGD implementation")
    ap.add_argument("--memory-format", default="contiguous", choices=("contiguous", "channels_last"),
                    help="--workload train: memory format of the stock convolution stack (NCHW as the reference, or NHWC)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="override the workload's batch size (experiments)")
    ap.add_argument("--graph", choices=("on", "off"), default="off",
                    help="replay the step as one HIP graph (falls back to eager launches if the capture fails).  Off by "
                         "default: with three launches per step the replay measured 7 %% slower than eager launches.  "
                         "--workload train (one GPU): the whole optimiser step, ~790 kernels, as one graph: 28.0 vs 30.2 ms")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the forward+backward leg")
    ap.add_argument("--no-per-step-leg", action="store_true", help="skip the S=1 schedule leg (profiling runs)")
    ap.add_argument("--no-sparse-leg", action="store_true", help="skip the sparse-depth variant leg (same workload with a 500-sample sparse depth)")
    ap.add_argument("--no-stock-ops-leg", action="store_true", help="skip the stock-PyTorch-ops baseline on the same GPU")
    ap.add_argument("--cold-sets", type=int, default=8,
                    help="extra leg: rotate over this many input sets (SURVEY.md 8d: >= 8 sets, > 256 MiB in total); 0/1 disables it")
    ap.add_argument("--prewarm-s", type=float, default=0.5, help="untimed steady-state pre-warm-up before the W warm-up steps")
    ap.add_argument("--conv-autotune", choices=("on", "off"), default="off",
                    help="--workload train: torch.backends.cudnn.benchmark (MIOpen find mode) as the reference's main.py:37; "
                         "slow first steps (minutes)")
    ap.add_argument("--no-metrics", action="store_true", help="leave the depth-metrics reduction out of the step")
    ap.add_argument("--affinity-channels", type=int, default=8,
                    help="--workload train: filters of the affinity head the forward runs (8 = what the CSPN module reads, round 6; 12 = the reference's tensor)")
    ap.add_argument("--infer-batch", type=int, default=0,
                    help="--workload train: also time an inference-only UNet + CSPN step (eval mode, no_grad) at this batch size")
    # round 5: the default is OFF — the training-shaped leg is GPU-bound since the host path through CSPN3Function was cut from 217 to
    # 181 us per pass (it no longer matters where the scheduler puts the threads: 217.3 us unbound against 220.2 bound, headline 514.1 k
    # against 514.4 k maps/s at 200 steps, profiles/r05_bench_default_bind_{off,auto}.json), and a binding the harness applies is a
    # property of the harness, not of the product
    ap.add_argument("--cpu-bind", choices=("auto", "off"), default="off",
                    help="off (default): leave the affinity mask alone; auto: bind the rank's host threads to four adjacent cores of the allowed set (see bind_cpus)")
    args = ap.parse_args()
    cpu_bound = bind_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))),
                          args.cpu_bind)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP engine has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and args.backend == "nccl":
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, n_dev))
    torch.cuda.set_device(local_rank % n_dev)           # (gloo dry runs may oversubscribe one GPU)
    device = torch.device("cuda", local_rank % n_dev)
    # CSPN_BENCH_FORCE_DIST=1 (with torchrun's environment): initialise the process group even at world size 1, so that the
    # N > 1 code path — RCCL init with a device id, barrier, all_reduce of the elapsed time, the metrics all-gather — runs on a
    # one-GPU box exactly as the driver's N > 1 launch line would run it
    force_dist = os.environ.get("CSPN_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    wl = dict(WORKLOADS[args.workload])
    if args.batch > 0:
        wl["B"] = args.batch
        wl["name"] += " [batch overridden to %d]" % args.batch
    if world > n_dev:
        F.set_resident("off")      # several ranks share one GPU (gloo dry run): resident launches must own the device
    if args.workload == "train":
        return run_train(args, wl, world, rank, local_rank, device)
    strong = args.workload == "kitti"
    plan = parse_plan(args.plan)
    K, T = wl["K"], wl["T"]
    if strong:                                        # config 4: ONE batch of 8, sharded across the ranks (contiguous chunks)
        lo, hi = pkg.evaluation.shard_bounds(wl["B"], rank, world)
        B_local = hi - lo
        full = make_inputs(wl, wl["B"], device, seed=1234, sparse=args.sparse)      # every rank regenerates the same batch
        lo_ = min(lo, wl["B"] - 1)                     # a rank beyond the batch keeps one frame so its launches are valid; it counts 0
        g, d, s, target = (None if t is None else t[lo_:max(hi, lo_ + 1)].contiguous() for t in full)
        del full
    else:
        B_local = wl["B"]
        g, d, s, target = make_inputs(wl, B_local, device, seed=1234 + rank, sparse=args.sparse)
    if K == 3:
        module = pkg.CSPN_new.AffinityPropagate(T, 3, plan=plan)
        run = lambda: module(g, d, s)                                  # noqa: E731
    else:
        module = pkg.CSPN_ours.AffinityPropagate(T, plan=plan, state_dtype=None)
        run = lambda: module(d, g, s)                                  # noqa: E731
    if plan == "auto":        # host-side autotuner: time the candidate plans once on this problem, then pin the winner
        with torch.no_grad():
            if K == 3:
                w_, _, _ = F.cspn3_prepare(g)
            else:
                w_, _ = F.pac_prepare(g)
            sp_ = None if s is None else s[:, 0].contiguous()
            plan = F.autotune_plan(w_, d[:, 0].contiguous(), sp_, K, T,
                                   F.BLEND_SPARSE if s is not None else F.BLEND_NONE, guidance=g if K == 3 else None,
                                   score=None if (args.no_metrics or K != 3) else (
                                       target[:, 0].contiguous(), pkg.evaluation.new_accumulator(device)))
            del w_
        module.plan = plan
    w_torch_dtype = torch.float16 if wl["dtype"] == "f16" else torch.float32
    eff_plan = F.resolve_plan(K, B_local, wl["H"], wl["W"], T, False,
                              plan if K == 3 else F.dtype_default_plan(K, w_torch_dtype, plan))
    # which schedule will the step run?  (the weight-resident single launch serves the no-grad 3x3 calls it fits)
    res_plan = None
    if K == 3 and B_local > 0:                              # (also under --graph on: the capture records the launch + a flag memset)
        res_plan = F.resident_supported(g, d[:, 0], None if s is None else s[:, 0], T, plan,
                                        None if args.no_metrics else target[:, 0])
    if K != 3 and B_local > 0 and plan is None:             # K x K: weight-resident launches for fp16 guidance (cspnk_forward_resident)
        res_plan = F.pac_resident_supported(g, d[:, 0].contiguous(), None if s is None else s[:, 0].contiguous(), T, plan,
                                            None if args.no_metrics else target[:, 0].contiguous())
    dot2 = False
    if res_plan is not None:
        eff_plan = dict(res_plan, schedule="resident", steps_per_launch=T)
        eff_plan.pop("debug_stamps", None)
        # K = 5 with fp16 guidance and fp16 planes (config 3): the dot-product kernel refines the chunks of the batch back to back
        # in ONE launch (csrc/cspnk_d2.hip) unless CSPN_KRES_STEP=fma pins the FMA kernel
        dot2 = (K == 5 and g.dtype == torch.float16 and d.dtype == torch.float16 and res_plan["quads_per_thread"] == 1
                and F._KRES_STEP_FORM != F.STEP_FMA)
        if dot2:
            eff_plan = dict(eff_plan, step_form="dot2 (v_dot2_f32_f16 on fp16 state pairs, state rounded to half every step)",
                            rounds_per_launch=res_plan["launches"], launches=1)
    else:
        eff_plan = dict(eff_plan, schedule="multi-launch")
    sums = pkg.evaluation.new_accumulator(device)

    if K == 3:
        run_scored = lambda acc: module.forward_scored(g, d, s, target, acc)       # noqa: E731
    else:
        run_scored = lambda acc: module.forward_scored(d, g, s, target, acc)       # noqa: E731

    def step():
        # the eval step of the pipeline: refine the batch and score it (metrics fused into the last launch)
        return run() if args.no_metrics else run_scored(sums)

    use_graph = args.graph == "on"
    if use_graph:
        # replay the whole step as one HIP graph (inputs are already resident: no per-step copies); the launches are the
        # same kernels with the same arguments, only the per-launch dispatch gaps shrink
        eager_step = step
        try:
            with torch.no_grad():
                graphed = pkg.graphs.GraphedForward(lambda: eager_step())
        except Exception as e:                                     # noqa: BLE001  (capture unsupported here: run eager)
            if rank == 0:
                print("bench: HIP-graph capture failed (%s); running eager launches" % (str(e).splitlines()[0][:120],),
                      file=sys.stderr)
            use_graph = False
    if use_graph:
        launches_per_step = -(-T // max(eff_plan["steps_per_launch"], 1))

        def step():                                                # noqa: F811
            log = F._EVENT_LOG                                     # HIP events around the replay, on the replay's stream
            if log is None:
                return graphed(copy_inputs=False)
            ev0, ev1 = log.pair()
            ev0.record()
            out = graphed(copy_inputs=False)
            ev1.record()
            log.append((ev0, ev1, launches_per_step, eff_plan["steps_per_launch"]))
            return out

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    gather = lambda acc: pkg.evaluation.all_gather_metric_sums(acc, force_collective=force_dist)     # noqa: E731

    with torch.no_grad():
        # untimed pre-warm-up (on top of the W warm-up steps): lets the caching allocator, the lazily created
        # HIP modules and the GPU clocks settle, so that short runs (small K/W) measure the same steady state
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_s:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        gather(sums)                                  # untimed: first use loads the reduction kernels / sets up RCCL
        sums.zero_()
        # Rehearsal of the timed region's own sequence (fence -> instrumented steps -> gather -> fence), untimed: the first
        # step issued right after a device-wide fence costs the host 90-240 us ONCE per process (lazy runtime state; 30 us
        # from the second time on) — a fifth of a 20-step region, nothing of a 200-step one
        F.set_event_log(F.EventLog(2, every=1))
        fence()
        step()
        step()
        gather(sums)
        fence()
        sums.zero_()
        # HIP events around every 10th step only (two records cost ~3 us of stream time each step: at every 4th they were 3 % of the
        # driver's 20-step region): still measured live inside the timed region, `launches_timed` says how many launches the
        # average is over
        events = F.EventLog(args.steps, every=10 if args.steps >= 20 else 1)
        F.set_event_log(events)
        fence()
        dbg_host = [] if os.environ.get("BENCH_DEBUG") else None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            if dbg_host is not None:
                dbg_host.append(time.perf_counter())
        t_loop = time.perf_counter()
        total, per_rank = gather(sums)                                 # the only collective: 10 float64 per rank
        t_gather = time.perf_counter()
        fence()
        t1 = time.perf_counter()
        if os.environ.get("BENCH_DEBUG") and rank == 0:
            print("debug: host loop %.2f ms, gather call %.2f ms, final fence %.2f ms" % (
                (t_loop - t0) * 1e3, (t_gather - t_loop) * 1e3, (t1 - t_gather) * 1e3), file=sys.stderr)
            print("debug: host us per step call: %s" % " ".join("%.0f" % ((b - a) * 1e6) for a, b in zip([t0] + dbg_host, dbg_host)),
                  file=sys.stderr)
            evs = list(events)
            if len(evs) > 1:
                print("debug: GPU us between sampled steps' starts: %s; sampled step durations: %s" % (
                    " ".join("%.0f" % (evs[i][0].elapsed_time(evs[i + 1][0]) * 1e3) for i in range(len(evs) - 1)),
                    " ".join("%.0f" % (e[0].elapsed_time(e[1]) * 1e3) for e in evs)), file=sys.stderr)
        F.set_event_log(None)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1 or force_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # ---- roofline of the dominant kernel (the propagation kernel), from HIP events on the launch stream
    per_fwd_us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1, _, _ in events)      # one entry per timed step

    def pct(q):
        return per_fwd_us[min(len(per_fwd_us) - 1, int(q * len(per_fwd_us)))] if per_fwd_us else None
    prop_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in events)
    n_launch = sum(n for _, _, n, _ in events)
    S = eff_plan["steps_per_launch"]
    esz = 2 if wl["dtype"] == "f16" else 4
    bytes_px_step = (K * K + 1) * esz + (2 * esz if args.sparse else 0)
    alg_bytes_per_launch = bytes_px_step * B_local * wl["H"] * wl["W"] * (T / max(1, -(-T // S)))
    avg_launch_s = (prop_ms / 1e3) / max(n_launch, 1)
    achieved = alg_bytes_per_launch / avg_launch_s / 1e9 if n_launch else 0.0
    # measured HBM traffic (rocprofv3 --pmc passes, tools/r02_profile_session.sh -> tools/pmc_traffic.py): NOT collected by this
    # run — counters need their own rocprofv3 passes — so the figures are quoted from the committed summary together with
    # the source digest of the kernels they were measured on (`traffic_stale` = the kernels changed since).
    pmc = load_pmc_traffic(args.workload + ("_sparse" if args.sparse else ""), "fused" if res_plan is not None else "multi",
                           eff_plan["launches"] if res_plan is not None else 1)
    if args.batch > 0 or world > 1:                     # the committed passes were measured on the workload's own batch on one GPU
        pmc = dict(pmc, step_bytes_per_launch=None, fused_bytes_per_forward=None, fused_per_launch=None, sq=None,
                   per_kernel_per_forward=None,
                   source=None if pmc["source"] is None else pmc["source"] + " [not applicable: batch overridden / sharded]")

    # ---- the north-star schedule: ONE launch per propagation step (S = 1), measured in the same process.  This is the
    # HBM-bound kernel the contract's `roofline` block describes: its algorithmic bytes ARE its HBM traffic.
    per_step = None
    if rank == 0 and not args.no_per_step_leg:
        with torch.no_grad():
            if K == 3:
                w1, _, _ = F.cspn3_prepare(g)
                d1 = d[:, 0].contiguous()
            else:
                w1, _ = F.pac_prepare(g)
                d1 = d[:, 0].contiguous()
            s1 = None if s is None else s[:, 0].contiguous()
            bl1 = F.BLEND_SPARSE if s is not None else F.BLEND_NONE
            best_us, p1 = float("inf"), None
            for cand in F.candidate_plans(K, wl["H"], wl["W"], 1):       # host-side autotuner restricted to S = 1
                if cand["steps_per_launch"] != 1:
                    continue
                try:
                    F.propagate(w1, d1, s1, K, T, bl1, plan=cand)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        F.propagate(w1, d1, s1, K, T, bl1, plan=cand)
                    e1.record()
                    e1.synchronize()
                except RuntimeError:
                    continue
                if e0.elapsed_time(e1) < best_us:
                    best_us, p1 = e0.elapsed_time(e1), cand
        n1 = max(10, args.steps // 4)
        ev1 = F.EventLog(n1)
        with torch.no_grad():
            for _ in range(5):
                F.propagate(w1, d1, s1, K, T, bl1, plan=p1)
            F.set_event_log(ev1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n1):              # one forward of the per-step schedule = prepare + T one-step launches
                if K == 3:
                    w1, _, _ = F.cspn3_prepare(g)
                else:
                    w1, _ = F.pac_prepare(g)
                F.propagate(w1, d1, s1, K, T, bl1, plan=p1)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t0
            F.set_event_log(None)
        del w1
        # ... and the same kernel where the Infinity Cache cannot help (SURVEY.md 8d asks for both): two input sets of
        # `Bc` frames each, every launch's own bytes >= 300 MB (> the 256 MiB cache), launches alternating between the
        # sets so that nothing a launch reads or writes was touched by the launch before it
        cold1 = None
        try:
            px = wl["H"] * wl["W"]
            Bc = max(B_local, -(-300_000_000 // (bytes_px_step * px)))
            with torch.no_grad():
                csets = []
                for i in range(2):
                    gc, dc, sc, _ = make_inputs(wl, Bc, device, seed=777 + i, sparse=args.sparse)
                    wc = (F.cspn3_prepare(gc)[0] if K == 3 else F.pac_prepare(gc)[0])
                    csets.append((wc, dc[:, 0].contiguous(), None if sc is None else sc[:, 0].contiguous()))
                    del gc, dc, sc
                best_c, pc = float("inf"), None
                for cand in F.candidate_plans(K, wl["H"], wl["W"], 1):
                    if cand["steps_per_launch"] != 1:
                        continue
                    try:
                        for wc, dc, sc in csets:
                            F.propagate(wc, dc, sc, K, 1, bl1, plan=cand)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(3):
                            for wc, dc, sc in csets:
                                F.propagate(wc, dc, sc, K, 1, bl1, plan=cand)
                        e1.record()
                        e1.synchronize()
                    except RuntimeError:
                        continue
                    if e0.elapsed_time(e1) < best_c:
                        best_c, pc = e0.elapsed_time(e1), cand
                nrep = 20
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(nrep):
                    for wc, dc, sc in csets:
                        F.propagate(wc, dc, sc, K, 1, bl1, plan=pc)
                e1.record()
                e1.synchronize()
                us_c = e0.elapsed_time(e1) * 1e3 / (2 * nrep)
            algc = bytes_px_step * Bc * px
            cold1 = {"batch": Bc, "input_sets": 2, "bytes_per_launch": algc, "footprint_MB": 2 * algc / 1e6,
                     "avg_launch_us": us_c, "achieved": algc / us_c / 1e3, "frac": algc / us_c / 1e3 / HBM_PEAK_GBS, "plan": pc,
                     "note": "one S=1 launch per call, alternating between two input sets (launch gaps included)"}
            del csets
        except RuntimeError as e:                                      # noqa: BLE001
            cold1 = {"error": str(e).splitlines()[0][:160]}
        ms1 = sum(e0.elapsed_time(e1) for e0, e1, _, _ in ev1)
        nl1 = sum(n for _, _, n, _ in ev1)
        alg1 = bytes_px_step * B_local * wl["H"] * wl["W"]
        a1 = alg1 / (ms1 / 1e3 / nl1) / 1e9
        per_step = {"bound": "hbm", "kernel": "cspn_prop_fused<%d,...> S=1 (one launch per propagation step)" % K,
                    "achieved": a1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a1 / HBM_PEAK_GBS,
                    "traffic": pmc["step_bytes_per_launch"], "traffic_source": pmc["source"],
                    "traffic_stale": pmc["stale"],
                    "algorithmic_bytes_per_launch": alg1, "bytes_per_px_step": bytes_px_step,
                    "avg_launch_us": ms1 * 1e3 / nl1, "launches_timed": nl1, "steps_per_launch": 1,
                    "maps_per_s_forward_only": B_local * n1 / dt1, "plan": p1,
                    "frac_cache_cold": None if not cold1 or "frac" not in cold1 else cold1["frac"], "cache_cold": cold1,
                    "note": "HIP events on the launch stream around each T-launch loop (launch gaps included) / T; "
                            "SURVEY.md 8(d): (K^2+1)*sizeof(T) bytes per pixel per step"}

    # ---- the default (temporally blocked) schedule, the one `value` is measured on: S steps per launch keep the depth
    # tile in LDS, so a launch moves FEWER HBM bytes than S x the per-step algorithmic bytes — HBM is then not its bound
    # and algorithmic bytes / time is an EFFECTIVE rate, not a roofline fraction.
    launches_fwd = -(-T // max(S, 1))
    esz_g = esz
    # what a forward cannot avoid moving: K^2-1 guidance planes + coarse depth in, refined depth out (+ the sparse
    # plane, + the target plane the fused metrics read)
    compulsory = B_local * wl["H"] * wl["W"] * ((K * K - 1) * esz_g + 2 * esz + (esz if args.sparse else 0) +
                                                (0 if args.no_metrics else esz))
    fused = {"kernel": (("cspn3_resident<%d,...>" if K == 3 else ("cspnk_d2<...> (%d oct per thread)" if dot2 else "cspnk_resident<%d,%%d,...>" % K)) % eff_plan["quads_per_thread"] +
                        " (%d launch(es) of whole images, weights resident in VGPRs for all %d steps, %d-step phases)" % (
                            eff_plan["launches"], T, eff_plan["steps_per_phase"])) if res_plan is not None else
             "cspn_prop_fused<%d,...> S=%d (%d launches per forward)" % (K, S, launches_fwd),
             "bound": "valu+lds (temporal blocking: the step loop runs out of LDS/VGPRs, HBM traffic is below the "
                      "algorithmic bytes)" if S > 1 else "hbm",
             "steps_per_launch": S, "launches_per_forward": launches_fwd,
             "avg_launch_us": avg_launch_s * 1e6, "launches_timed": n_launch,
             "algorithmic_bytes_per_launch": alg_bytes_per_launch,
             "effective_algorithmic_GBs": achieved,
             "effective_over_hbm_peak": achieved / HBM_PEAK_GBS,
             "compulsory_bytes_per_forward": compulsory,
             "hbm_traffic_bytes_per_forward": pmc["fused_bytes_per_forward"],
             "hbm_traffic_per_launch": pmc["fused_per_launch"],
             "hbm_traffic_per_kernel_bytes_x_launches_per_forward": pmc["per_kernel_per_forward"],
             "traffic_source": pmc["source"], "traffic_stale": pmc["stale"],
             "propagation_us_per_forward_p10_p50_p90": [pct(0.10), pct(0.50), pct(0.90)]}
    if pmc["fused_bytes_per_forward"] and n_launch:
        # the traffic is that of the WHOLE forward (prepare / softmax pass, every propagation launch, metrics), so it is
        # priced against the whole step's time (max over ranks of the timed region / steps), not the propagation launches'
        fwd_s = elapsed / args.steps
        fused["hbm_GBs_on_measured_traffic"] = pmc["fused_bytes_per_forward"] / fwd_s / 1e9
        fused["frac_hbm_on_measured_traffic"] = fused["hbm_GBs_on_measured_traffic"] / HBM_PEAK_GBS
        fused["traffic_over_compulsory"] = pmc["fused_bytes_per_forward"] / compulsory
    if pmc.get("sq"):
        fused["sq_counters"] = pmc["sq"]

    # ---- cache-cold leg: rotate over input sets whose footprint exceeds the 256 MiB Infinity
    # Cache, so that no forward finds its guidance / depth / target resident from the previous one
    cold = None
    if rank == 0 and args.cold_sets > 1:
        sets = [(g, d, s, target)] + [make_inputs(wl, max(B_local, 1), device, seed=99 + i, sparse=args.sparse)
                                      for i in range(args.cold_sets - 1)]
        foot = sum(sum(t.numel() * t.element_size() for t in st if t is not None) for st in sets)
        acc2 = pkg.evaluation.new_accumulator(device)

        def cold_step(i):
            gg, dd, ss, tt = sets[i % len(sets)]
            if args.no_metrics:
                module(gg, dd, ss) if K == 3 else module(dd, gg, ss)
            elif K == 3:
                module.forward_scored(gg, dd, ss, tt, acc2)
            else:
                module.forward_scored(dd, gg, ss, tt, acc2)

        with torch.no_grad():
            for i in range(2 * len(sets)):
                cold_step(i)
            torch.cuda.synchronize()
            nc = max(len(sets) * 4, args.steps // 2)
            t0c = time.perf_counter()
            for i in range(nc):
                cold_step(i)
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t0c
        cold = {"value": B_local * nc / dtc, "unit": "depth-maps/s", "ms_per_step": dtc / nc * 1e3,
                "input_sets": len(sets), "footprint_MB": foot / 1e6, "steps": nc}
        del sets

    # ---- the sparse-depth variant of the same workload, in the same process (VERDICT r5 missing #5: the reference's call site ALWAYS
    # passes sparse_depth — network/unet_cspn_nyu.py:362,386 — and SURVEY.md 8(d) asks for "with and without"): the scored forward on the
    # default schedule, and the S = 1 roofline kernel with the blend fused at (K^2 + 3) * sizeof(T) bytes per pixel and step
    sparse_variant = None
    if rank == 0 and not args.sparse and not args.no_sparse_leg and B_local > 0:
        try:
            gs, ds, ss, ts = make_inputs(wl, B_local, device, seed=1234 + rank, sparse=True)      # the same g / d / target + 500 samples
            acc_s = pkg.evaluation.new_accumulator(device)
            step_s = (lambda: module.forward_scored(gs, ds, ss, ts, acc_s)) if K == 3 else (lambda: module.forward_scored(ds, gs, ss, ts, acc_s))
            with torch.no_grad():
                for _ in range(20):
                    step_s() if not args.no_metrics else (module(gs, ds, ss) if K == 3 else module(ds, gs, ss))
                torch.cuda.synchronize()
                ns = max(args.steps, 100)
                t0s = time.perf_counter()
                for _ in range(ns):
                    step_s() if not args.no_metrics else (module(gs, ds, ss) if K == 3 else module(ds, gs, ss))
                torch.cuda.synchronize()
                dts = (time.perf_counter() - t0s) / ns
            F.ensure_resident_ok(device)       # (rank 0 only: no collective here)
            bps = (K * K + 1) * esz + 2 * esz
            sparse_variant = {"value": B_local / dts, "unit": "depth-maps/s", "ms_per_step": dts * 1e3, "steps": ns,
                              "bytes_per_px_step": bps, "sparse_samples_per_frame": 500,
                              "metrics_check": {k: v for k, v in pkg.evaluation.finalize_metrics(acc_s.sum(0).cpu()).items()
                                                if k in ("rmse", "count")}}
            if not args.no_per_step_leg:
                with torch.no_grad():
                    ws_ = (F.cspn3_prepare(gs)[0] if K == 3 else F.pac_prepare(gs)[0])
                    d1s, s1s = ds[:, 0].contiguous(), ss[:, 0].contiguous()
                    ps = None if per_step is None else per_step.get("plan")
                    for _ in range(3):
                        F.propagate(ws_, d1s, s1s, K, T, F.BLEND_SPARSE, plan=ps)
                    nfs = 10
                    evs = F.EventLog(nfs)
                    F.set_event_log(evs)
                    torch.cuda.synchronize()
                    for _ in range(nfs):
                        F.propagate(ws_, d1s, s1s, K, T, F.BLEND_SPARSE, plan=ps)
                    torch.cuda.synchronize()
                    F.set_event_log(None)
                    mss = sum(e0.elapsed_time(e1) for e0, e1, _, _ in evs)
                    nls = sum(n for _, _, n, _ in evs)
                    algs = bps * B_local * wl["H"] * wl["W"]
                    a_s = algs / (mss / 1e3 / nls) / 1e9
                    sparse_variant["roofline"] = {"bound": "hbm", "kernel": "cspn_prop_fused<%d,...> S=1 with the sparse blend fused" % K,
                                                  "achieved": a_s, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_s / HBM_PEAK_GBS,
                                                  "algorithmic_bytes_per_launch": algs, "avg_launch_us": mss * 1e3 / nls,
                                                  "launches_timed": nls, "steps_per_launch": 1, "traffic": None}
                    del ws_
            del gs, ds, ss, ts
        except RuntimeError as e:                                      # noqa: BLE001
            sparse_variant = {"error": str(e).splitlines()[0][:200]}
            F.set_event_log(None)

    # ---- the reference's own GPU path as a same-box baseline: the stock-ops port on this GPU (outside the timed region)
    stock = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_stock_ops_leg and B_local > 0:
        try:
            with torch.no_grad():
                hip_ref = (module(g, d, s) if K == 3 else module(d, g, s))
            stock = stock_ops_same_gpu(wl, g, d, s, device, hip_out=hip_ref)
            del hip_ref
        except Exception as e:                                         # noqa: BLE001
            stock = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()

    # ---- training-shaped use of the same module (config 5's CSPN share): forward with history + hand-written backward
    train = None
    if rank == 0 and not args.no_train_leg:
        gt = g.detach().clone().requires_grad_(True)
        dt_ = d.detach().clone().requires_grad_(True)
        cot = torch.randn_like(d)

        def fwd_bwd():
            gt.grad = None
            dt_.grad = None
            out = module(gt, dt_, s) if K == 3 else module(dt_, gt, s)
            out.backward(cot.to(out.dtype))

        # (10 untimed + >= 100 timed passes; 30 until late in round 5: the pipeline fill and the closing synchronize were ~3 % of them.
        #  Since round 5 the step is GPU-bound at configs 2 / 3 — lean host path, device-side guard: no host wait at the end of backward —
        #  and host-bound, hence placement-sensitive, at shard sizes: see bind_cpus)
        for _ in range(10):
            fwd_bwd()
        torch.cuda.synchronize()
        nt = max(100, args.steps // 2)
        t0t = time.perf_counter()
        for _ in range(nt):
            fwd_bwd()
        torch.cuda.synchronize()
        dtt = (time.perf_counter() - t0t) / nt
        train = {"fwd_bwd_us": dtt * 1e6, "maps_per_s": B_local / dtt, "steps": nt,
                 "note": "CSPN module only (forward keeping T depth planes + reverse sweep + fused backward tail)"}
        tb = None if args.batch > 0 else load_train_traffic(args.workload + ("_sparse" if args.sparse else ""))
        if tb is not None:           # HBM roofline of the training-shaped step on its MEASURED traffic (PMC passes, committed file)
            train["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                 "hbm_traffic_bytes_per_step": tb["bytes_per_step"],
                                 "achieved": tb["bytes_per_step"] / dtt / 1e9,
                                 "frac": tb["bytes_per_step"] / dtt / 1e9 / HBM_PEAK_GBS,
                                 "per_kernel_bytes_x_launches": tb["per_kernel"], "traffic_source": tb["source"],
                                 "traffic_stale": tb["stale"],
                                 "note": "achieved = measured HBM bytes of one forward+backward / its wall time (launch gaps and "
                                         "autograd host time included)"}
        del gt, dt_, cot

    # ---- what a plain device copy achieves on this GPU (fraction of achievable, next to nominal)
    copy_gbs = None
    if rank == 0:
        src_c = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=device)      # 256 MiB, beyond the L2
        dst_c = torch.empty_like(src_c)
        for _ in range(3):
            dst_c.copy_(src_c)
        e0c, e1c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0c.record()
        for _ in range(10):
            dst_c.copy_(src_c)
        e1c.record()
        e1c.synchronize()
        copy_gbs = 10 * 2 * src_c.numel() * 4 / (e0c.elapsed_time(e1c) / 1e3) / 1e9
        del src_c, dst_c

    maps_total = (wl["B"] if strong else wl["B"] * world) * args.steps
    if rank == 0:
        res = {
            "metric": "depth-maps/sec at 304x228 b=24, 24 CSPN iters" if args.workload == "nyu"
                      else "depth-maps/sec (%s)" % wl["name"],
            "value": maps_total / elapsed,
            "unit": "depth-maps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": wl["dtype"],
            "dtype_note": ("fp32 arithmetic and storage; parity bar 1e-5 relative / RMSE 1e-4 against the reference (north_star)"
                           if wl["dtype"] == "f32" else
                           "fp16 storage (guidance, taps, depth planes), fp32 accumulation inside a step; north_star states 1e-5 for fp32 "
                           "only, so this configuration's bar is anchored on the REFERENCE's own half-precision run (golden G15, round 6: "
                           "CSPN_ours under the default dtype float16 is 8.1e-4 x max / 2.4e-4 x max RMSE away from the fp32 result on "
                           "fp16-rounded inputs; this path must be no further: measured 7.8e-4 / 2.3e-4, tests/test_hip_kres.py); the "
                           "full-batch test keeps the older 8e-3 / 3e-3 guard rail (tests/test_hip_production.py)"),
            "data": "synthetic (guidance~N(0,1), coarse~U(0,10)m%s; inputs resident in HBM)" % (
                ", 500-sample sparse depth" if args.sparse else ""),
            "config": {"workload": wl["name"], "batch_per_gpu": B_local, "H": wl["H"], "W": wl["W"], "K": K,
                       "prop_time": T, "guidance_channels": wl["C"], "sparse": bool(args.sparse),
                       "step": "prepare + %d propagation steps%s" % (T, "" if args.no_metrics else " + depth metrics (fused into the last launch)"),
                       "plan": eff_plan, "hip_graph": bool(use_graph), "parallelism": "batch-shard x%d, metrics all-gather" % world,
                       "host_cpu_bind": cpu_bound},
            "roofline": per_step if per_step is not None else (
                dict(fused, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=None)
                if S == 1 else None),
            "default_schedule": fused,
            "metrics_check": {k: v for k, v in pkg.evaluation.finalize_metrics(total.cpu()).items()
                              if k in ("rmse", "absrel", "delta1", "count")},
        }
        if copy_gbs and res["roofline"] is not None:
            res["roofline"]["device_copy_GBs"] = copy_gbs
            res["roofline"]["frac_of_device_copy"] = res["roofline"]["achieved"] / copy_gbs
        if cold is not None:
            res["cache_cold"] = cold
        if sparse_variant is not None:
            res["sparse_variant"] = sparse_variant
        if stock is not None:
            res["stock_ops_same_gpu"] = stock
        if train is not None:
            res["training_step"] = train
        if world == 1 and not args.no_cpu_baseline:
            try:
                with full_affinity():
                    res["cpu_baseline"] = cpu_baseline(wl)
            except Exception as e:  # pragma: no cover
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
