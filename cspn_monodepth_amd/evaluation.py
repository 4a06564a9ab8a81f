"""Depth metrics on device + the batch-shard / metrics all-gather used for multi-GPU inference.

Reference formulas: libs/metrics.py:49-83 (Result.evaluate) and the on-device twin
network/libs/base/base_model.py:28-73, whose 10-vector order (irmse, imae, mse, rmse, mae, absrel,
lg10, delta1, delta2, delta3) is kept.  The reference averages the per-GPU vectors with an
in-process Reduce (network/libs/base/encoding.py:264-276), which is only right for equal valid-pixel
counts; here every rank contributes additive masked *sums* + the count, gathered with one
torch.distributed all_gather (RCCL over xGMI when the backend is "nccl"), then finalised.

Deviation to know about: `finalize_metrics` is pixel-weighted over everything accumulated (one square root at the end).
The reference's single-GPU eval loop prints the batch-size-weighted mean of PER-BATCH metrics (Result.evaluate +
AverageMeter, libs/metrics.py:49-127) — `BatchAverageMeter` below reproduces exactly that for comparable numbers.
"""
import ctypes
import math

import torch
import torch.distributed as dist

from . import _lib
from . import functional as _F

METRIC_NAMES = ("irmse", "imae", "mse", "rmse", "mae", "absrel", "lg10", "delta1", "delta2", "delta3")
N_SUMS = 10   # {inv^2, inv, diff^2, diff, diff/t, |dlog10|, #<1.25, #<1.25^2, #<1.25^3, n}
N_SLOTS = 1024  # accumulator rows: >= workgroups per launch, so no two blocks contend on one fp64 atomic address


def new_accumulator(device):
    """Zeroed [N_SLOTS, 10] float64 accumulator for metric_sums(..., out=acc) in a loop over batches."""
    return torch.zeros((N_SLOTS, N_SUMS), dtype=torch.float64, device=device)


def shard_bounds(n_items, rank, world_size):
    """Contiguous batch chunks: rank r gets [lo, hi); remainders go to the low ranks."""
    base, rem = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def metric_sums(pred, target, out=None):
    """Masked sums over target > 0 (one fused HIP reduction kernel).

    Without `out`: returns the float64[10] sums.  With `out` (from new_accumulator): accumulates into its
    rows and returns it, so a loop over batches needs neither a host sync nor an extra reduction launch."""
    if not (pred.is_cuda and target.is_cuda):
        raise RuntimeError("metric_sums: tensors must live on a ROCm device (no CPU implementation here)")
    if pred.shape != target.shape or pred.dtype != target.dtype:
        raise ValueError("pred / target must have the same shape and dtype")
    p, t = pred.contiguous(), target.contiguous()
    acc = new_accumulator(p.device) if out is None else out
    if acc.dtype != torch.float64 or acc.dim() != 2 or acc.shape[1] != N_SUMS or not acc.is_contiguous():
        raise ValueError("out must be a contiguous float64 [nslots, 10] tensor (evaluation.new_accumulator)")
    dt = _lib.CSPN_F32 if p.dtype == torch.float32 else _lib.CSPN_F16 if p.dtype == torch.float16 else None
    if dt is None:
        raise TypeError("metric_sums supports float32 / float16")
    with torch.cuda.device(p.device):
        ok = _lib.lib().cspn_metrics_accumulate(
            ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(t.data_ptr()), dt, p.numel(),
            ctypes.c_void_p(acc.data_ptr()), int(acc.shape[0]),
            ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
    _lib.check(ok, "cspn_metrics_accumulate")
    return acc.sum(0) if out is None else acc


def finalize_metrics(sums):
    """float64[10] sums -> dict of the reference's 10 metrics + 'count'."""
    # the numbers are about to be used on the host: a weight-resident launch that timed out is repaired here (or raised, when it
    # cannot be), not averaged in (a device tensor is waited for first; a host tensor has been through a synchronising copy already)
    if getattr(sums, "is_cuda", False):
        _F.ensure_resident_ok(sums.device)
    else:
        _F.check_resident_errors()
    if hasattr(sums, "dim") and sums.dim() == 2:
        sums = sums.sum(0)
    s = [float(v) for v in sums]
    n = s[9]
    if n <= 0:
        return dict({k: float("nan") for k in METRIC_NAMES}, count=0)
    mse = s[2] / n
    return dict(irmse=math.sqrt(s[0] / n), imae=s[1] / n, mse=mse, rmse=math.sqrt(mse), mae=s[3] / n,
                absrel=s[4] / n, lg10=s[5] / n, delta1=s[6] / n, delta2=s[7] / n, delta3=s[8] / n,
                count=int(round(n)))


class BatchAverageMeter(object):
    """The reference's averaging, for numbers comparable with its logs: `Result.evaluate` finalises the metrics PER BATCH
    (libs/metrics.py:49-83: rmse = sqrt of that batch's mse, irmse likewise) and `AverageMeter` averages those per-batch
    values weighted by the batch size n (libs/metrics.py:101-127, called with n = input.size(0) by the trainers).

    `finalize_metrics` over accumulated sums is the pixel-weighted GLOBAL figure instead (one sqrt over all pixels):
    the two agree for mse / mae / absrel / lg10 / delta only when every batch has the same valid-pixel count, and
    never exactly for rmse / irmse (mean of square roots != square root of the mean).  Use this class to reproduce the
    reference's printed averages; use the sums for the all-gathered multi-GPU figure.

        meter = BatchAverageMeter()
        for batch: meter.update(metric_sums(pred, target), n=pred.shape[0])      # one float64[10] per batch
        meter.average()  ->  dict of the 10 metrics (+ 'count' = number of samples, as AverageMeter.count)"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.count = 0.0
        self.sums = {k: 0.0 for k in METRIC_NAMES}

    def update(self, batch_sums, n=1):
        # (a batch without valid pixels records NaN, as the reference's Result.evaluate would)
        fin = finalize_metrics(batch_sums.cpu() if hasattr(batch_sums, "cpu") else batch_sums)
        self.count += n
        for k in METRIC_NAMES:
            self.sums[k] += n * fin[k]
        return fin

    def average(self):
        if self.count <= 0:
            return dict({k: float("nan") for k in METRIC_NAMES}, count=0)
        return dict({k: self.sums[k] / self.count for k in METRIC_NAMES}, count=self.count)


def all_gather_metric_sums(sums, group=None, force_collective=False):
    """All-gather the per-rank sums (world x 10 float64) and add them.  Works with gloo (CPU) and nccl/RCCL.

    This is where an evaluation loop hands its numbers on, so it first waits for the weight-resident launches that
    produced them; one that timed out is re-run on the multi-launch schedule and its metric sums are corrected
    (functional.ensure_resident_ok) — also for the last batch of the loop, which no later launch would check.  Returns (total [10], per_rank [world, 10]), independent tensors."""
    initialised = dist.is_available() and dist.is_initialized()
    if not initialised or (dist.get_world_size(group) == 1 and not force_collective):
        # one rank: nothing to gather (force_collective: run the collective anyway — the world-size-1 RCCL test and bench.py's
        # forced group).  The copies are enqueued BEHIND the launches first and the wait + check comes after: the same
        # guarantee (nothing is returned from a timed-out launch), without two kernel launches into an idle queue
        n0 = _F.resident_fallbacks()
        total = sums.sum(0) if sums.dim() == 2 else sums.clone()
        per_rank = total.clone().unsqueeze(0)
        if sums.is_cuda:
            _F.ensure_resident_ok(sums.device)
            if _F.resident_fallbacks() != n0:          # a timed-out launch was repaired (its sums corrected) after the copies
                total = sums.sum(0) if sums.dim() == 2 else sums.clone()
                per_rank = total.clone().unsqueeze(0)
        return total, per_rank
    if sums.is_cuda:
        _F.ensure_resident_ok(sums.device)
    if sums.dim() == 2:
        sums = sums.sum(0)
    world = dist.get_world_size(group)
    src = sums.contiguous()
    if sums.is_cuda and dist.get_backend(group) == "gloo":
        src = src.cpu()                                   # gloo dry runs: gather through host memory
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    stacked = torch.stack(parts, 0).to(sums.device)
    return stacked.sum(0), stacked
