#!/usr/bin/env python3
"""Worker of tests/test_distributed_gpu.py — launched with `python -m torch.distributed.run --nproc-per-node N`.

Every rank regenerates the SAME seeded batch, refines its contiguous shard with the HIP module (forward_scored: the
metrics fused into the last launch), and the per-rank metric sums are all-gathered (evaluation.all_gather_metric_sums).
Rank 0 then runs the unsharded forward on the same GPU and checks:
  * the shards cover the batch exactly once (contiguous [lo, hi) chunks);
  * the refined depth of every shard equals the corresponding slice of the unsharded forward BIT FOR BIT (the per-pixel
    arithmetic does not depend on the batch size or the launch plan);
  * the gathered metric sums equal the single-process ones (fp64 sums of fp32 per-wave partials; atomics may reorder
    the last bits: rtol 1e-12) and the oracle's on the oracle's refined depth (rtol 2e-5).
Backend gloo with every rank on the one visible GPU (the oversubscription branch of bench.py) or nccl with one GPU each."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import cspn_monodepth_amd as pkg
    from oracle import c_oracle, cspn_oracle as orc
    backend, B, H, W, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", local % n_dev)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    ev = pkg.evaluation
    g, d, s = c_oracle.synthetic_inputs(5, B, H, W, 12, 200)
    tgt = np.maximum(d + 0.1 * c_oracle.hash_normal(6, 9, d.shape), 0.0).astype(np.float32)
    lo, hi = ev.shard_bounds(B, rank, world)
    m = pkg.CSPN_new.AffinityPropagate(T, 3)
    acc = ev.new_accumulator(dev)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)        # noqa: E731
    out = torch.zeros((0, 1, H, W), device=dev)
    if hi > lo:
        with torch.no_grad():
            out = m.forward_scored(to(g[lo:hi]), to(d[lo:hi]), to(s[lo:hi]), to(tgt[lo:hi]), acc)
    total, stacked = ev.all_gather_metric_sums(acc)
    # gather the shards on rank 0 (verification only; the product path gathers just the 10 sums)
    bounds = [None] * world
    dist.all_gather_object(bounds, (lo, hi))
    parts = [None] * world
    dist.all_gather_object(parts, out.cpu().numpy())
    ok = True
    if rank == 0:
        assert bounds[0][0] == 0 and bounds[-1][1] == B and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1)), bounds
        with torch.no_grad():
            acc1 = ev.new_accumulator(dev)
            ref = m.forward_scored(to(g), to(d), to(s), to(tgt), acc1)
        refn = ref.cpu().numpy()
        cat = np.concatenate(parts, 0)
        assert cat.shape == refn.shape and np.array_equal(cat, refn), "sharded refined depth differs from the unsharded forward"
        assert np.allclose(total.cpu().numpy(), acc1.sum(0).cpu().numpy(), rtol=1e-12, atol=0)
        assert stacked.shape == (world, ev.N_SUMS)
        want = orc.metric_sums(c_oracle.cspn3_forward(g, d, s, T), tgt)
        assert np.allclose(total.cpu().numpy(), want, rtol=2e-5)
        print("SHARD_CHECK_OK world=%d backend=%s bounds=%s" % (world, backend, bounds), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
