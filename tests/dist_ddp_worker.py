#!/usr/bin/env python3
"""Worker of tests/test_distributed_gpu.py::test_ddp_training_step_two_ranks — launched with
`python -m torch.distributed.run --nproc-per-node 2`: BASELINE config 5's data-parallel branch (the re-hosted
unet_cspn_nyu topology under DistributedDataParallel + nn.SyncBatchNorm, HIP CSPN in forward and backward) with
world size 2.  Reference behaviour being replaced: libs/trainers/multi_gpu_trainer.py:32-37 (DataParallelModel /
DataParallelCriterion) and the sync-BN of network/libs/inplace_abn (functions.py:185-205, :271-274).

Backend gloo with both ranks on the one visible GPU (the box has one), or nccl with one GPU each.  Checks, all ranks:
  * SyncBatchNorm really synchronises: the batch-norm running means are equal across the ranks although their inputs differ;
  * DDP really averages: the gradients after a DDP backward equal the MEAN over the ranks of the gradients of the same
    forward/backward under `no_sync()` (each rank's own gradient), for the affinity head, the depth head and the stem;
  * the CSPN pair inside the model (guidance, coarse, sparse -> refined) matches the oracle, and its gradient flows;
  * after the optimiser steps the parameters are identical on both ranks and the loss is finite."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import cspn_monodepth_amd as pkg   # noqa: F401
    from cspn_monodepth_amd.network import unet_cspn_nyu
    from oracle import c_oracle
    backend, B = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    H, W = 228, 304
    torch.manual_seed(0)                                       # same initial weights on every rank (DDP broadcasts anyway)
    model = unet_cspn_nyu.resnet50(reference_state_dict=False).to(dev)
    model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    ddp = nn.parallel.DistributedDataParallel(model, device_ids=[dev.index] if backend == "nccl" else None)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    gen = torch.Generator(device=dev).manual_seed(77 + rank)  # DIFFERENT data per rank
    depth = torch.rand(B, 1, H, W, device=dev, generator=gen) * 9.5 + 0.5
    rgb = torch.rand(B, 3, H, W, device=dev, generator=gen)
    sparse = depth * (torch.rand(B, 1, H, W, device=dev, generator=gen) < 500.0 / (H * W))
    x = torch.cat([rgb, sparse], 1)
    target = torch.where(torch.rand(B, 1, H, W, device=dev, generator=gen) < 0.05, torch.zeros_like(depth), depth)

    captured = {}

    def hook(_m, inp, out):
        captured["in"] = [None if t is None else t.detach() for t in inp]
        captured["out"] = out.detach()
    model.post_process_layer.register_forward_hook(hook)

    def loss_of():
        pred = ddp(x)
        valid = target > 0
        return ((target - pred).abs() * valid).sum() / valid.sum()          # MaskedL1Loss (criteria.py:27-39)

    watched = {"affinity_head": model.gud_up_proj_layer6, "depth_head": model.gud_up_proj_layer5, "stem": model.conv1_1}
    wparams = {k: next(p for p in m.parameters()) for k, m in watched.items()}

    # 1. each rank's own gradient: same forward (SyncBN statistics are global either way), no gradient all-reduce
    opt.zero_grad(set_to_none=True)
    with ddp.no_sync():
        loss_of().backward()
    own = {k: p.grad.detach().clone() for k, p in wparams.items()}
    # 2. the DDP gradient of the same step
    opt.zero_grad(set_to_none=True)
    loss = loss_of()
    loss.backward()
    for k, p in wparams.items():
        parts = [torch.empty_like(own[k]) if backend == "nccl" else torch.empty_like(own[k]).cpu() for _ in range(world)]
        dist.all_gather(parts, own[k] if backend == "nccl" else own[k].cpu())
        mean = torch.stack([t.to(dev) for t in parts]).mean(0)
        scale = float(mean.abs().max())
        err = float((p.grad - mean).abs().max())
        # two backward passes of a 50-layer network through MIOpen's atomics-based weight gradients are not bit-reproducible:
        # measured run to run, relative to the largest entry: 9e-4 .. 7e-3 at the heads, 1.1e-2 at the stem (the far end of
        # the backward pass); the ranks' own gradients differ from the mean by far more (checked below)
        assert scale > 0 and err <= 3e-2 * scale, "DDP gradient of %s is not the mean of the ranks' gradients: %g vs scale %g" % (k, err, scale)
        if world > 1:       # the ranks' own gradients differ from the mean by far more than that noise: they saw different data
            assert float((own[k] - mean).abs().max()) > 4 * err + 1e-6 * scale, "ranks saw the same data? (%s)" % k
    # 3. the CSPN pair inside the model against the oracle
    gd, cd, sd = (t.cpu().numpy() for t in captured["in"])
    want = c_oracle.cspn3_forward(gd, cd, sd, 24)
    got = captured["out"].cpu().numpy()
    rel = float(np.abs(got - want).max() / max(1e-12, np.abs(want).max()))
    assert rel <= 1e-5, "CSPN pair inside the DDP model deviates from the oracle: %g" % rel
    assert float(wparams["affinity_head"].grad.abs().max()) > 0, "no gradient reached the affinity head through the HIP backward"
    # 4. optimiser steps; parameters and SyncBN statistics stay identical across ranks
    opt.step()
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss = loss_of()
        loss.backward()
        opt.step()
    assert bool(torch.isfinite(loss)), "loss became non-finite"
    digest = torch.stack([p.detach().double().sum() for p in model.parameters()] +
                         [b.detach().double().sum() for n, b in model.named_buffers() if n.endswith("running_mean")]).cpu()
    parts = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(parts, digest)
    for t in parts[1:]:
        assert torch.equal(t, parts[0]), "parameters / SyncBN statistics differ between the ranks after the optimiser steps"
    dist.barrier()
    if rank == 0:
        print("DDP_CHECK_OK world=%d loss=%.4f cspn_rel=%.2e" % (world, float(loss), rel), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
