"""world_size-2 gloo test of the N>1 path: contiguous batch shards, no collective inside the
propagation loop, one all-gather of the per-rank metric sums at the end.  Runs on CPU: the shard arithmetic,
the gather and the finalisation are host logic; the per-rank refinement itself is stood in for by the C
oracle here (the HIP path has no CPU implementation) — the -m gpu tests cover the kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, H, W, T, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cspn_monodepth_amd import evaluation as ev
    from oracle import c_oracle, cspn_oracle as orc
    g, d, s = c_oracle.synthetic_inputs(3, B, H, W, 12, 40)           # every rank regenerates the same batch
    target = np.maximum(d + 0.1 * c_oracle.hash_normal(4, 9, d.shape), 0.0).astype(np.float32)
    lo, hi = ev.shard_bounds(B, rank, world)
    out = c_oracle.cspn3_forward(g[lo:hi], d[lo:hi], s[lo:hi], T) if hi > lo else np.zeros((0, 1, H, W), np.float32)
    sums = torch.from_numpy(orc.metric_sums(out, target[lo:hi]))
    total, stacked = ev.all_gather_metric_sums(sums)
    assert stacked.shape == (world, ev.N_SUMS)
    assert torch.equal(stacked[rank], sums)
    np.save(os.path.join(outdir, "rank%d.npy" % rank), np.concatenate([total.numpy(), [lo, hi]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 2, 1])
def test_sharded_metrics_equal_single_process(tmp_path, B):
    world, H, W, T = 2, 20, 24, 6
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, H, W, T, str(tmp_path)), nprocs=world, join=True)
    from cspn_monodepth_amd import evaluation as ev
    from oracle import c_oracle, cspn_oracle as orc
    g, d, s = c_oracle.synthetic_inputs(3, B, H, W, 12, 40)
    target = np.maximum(d + 0.1 * c_oracle.hash_normal(4, 9, d.shape), 0.0).astype(np.float32)
    want = orc.metric_sums(c_oracle.cspn3_forward(g, d, s, T), target)
    res = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)]
    assert res[0][-2] == 0 and res[-1][-1] == B and res[0][-1] == res[1][-2]     # contiguous cover
    for r in res:
        assert np.allclose(r[:10], want, rtol=1e-12)                            # every rank holds the global sums
    fin = ev.finalize_metrics(torch.from_numpy(res[0][:10]))
    ref, n = orc.evaluate_metrics(c_oracle.cspn3_forward(g, d, s, T), target)
    assert fin["count"] == n
    for k, v in zip(ev.METRIC_NAMES, ref):
        assert np.isclose(fin[k], v, rtol=1e-6), k
