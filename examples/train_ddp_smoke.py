#!/usr/bin/env python3
"""Training-step plumbing for BASELINE config 5: a stock PyTorch-ROCm encoder/decoder with the reference's two
heads (1-channel coarse depth + bias-free 12-channel affinity head, network/unet_cspn_nyu.py:331-332, :383-386)
feeding the HIP CSPN module in forward AND backward, MaskedL1Loss (libs/criterion/criteria.py:27-39), SGD, and
DistributedDataParallel over RCCL when launched with torchrun:

    python examples/train_ddp_smoke.py                                   # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ddp_smoke.py

`--model resnet50` trains the re-hosted reference topology (cspn_monodepth_amd/network/unet_cspn_nyu.py: ResNet-50
encoder, guided up-projection decoder on the HIP un-pooling kernel, both heads — 218 M parameters in use); the default
`--model tiny` is a five-conv network with the same tensor contract at the CSPN boundary (guidance [B,12,H,W], coarse
[B,1,H,W], sparse = input[:,3:4]) for quick smoke runs.  `bench.py --workload train` is the measured version of the
resnet50 step.  Synthetic data (no dataset on the box).
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cspn_monodepth_amd.post_process import CSPN_new as post_process     # noqa: E402  (the one-line swap)


class TinyDepthNet(nn.Module):
    def __init__(self, width=32, prop_time=24):
        super().__init__()
        def block(i, o, s=1):
            return nn.Sequential(nn.Conv2d(i, o, 3, s, 1, bias=False), nn.BatchNorm2d(o), nn.ReLU(inplace=True))
        self.enc1, self.enc2, self.enc3 = block(4, width), block(width, 2 * width, 2), block(2 * width, 4 * width, 2)
        self.dec2, self.dec1 = block(4 * width + 2 * width, 2 * width), block(2 * width + width, 64)
        self.depth_head = nn.Conv2d(64, 1, 3, 1, 1, bias=False)            # gud_up_proj_layer5 analogue (:331)
        self.guidance_head = nn.Conv2d(64, 12, 3, 1, 1, bias=False)        # gud_up_proj_layer6 analogue (:332)
        self.post_process_layer = post_process.AffinityPropagate(prop_time, 3)   # unet_cspn_nyu.py:357-358

    def forward(self, x):
        sparse_depth = x[:, 3:4].clone()                                    # unet_cspn_nyu.py:362
        e1 = self.enc1(x); e2 = self.enc2(e1); e3 = self.enc3(e2)
        d2 = self.dec2(torch.cat([F.interpolate(e3, size=e2.shape[-2:], mode="nearest"), e2], 1))
        d1 = self.dec1(torch.cat([F.interpolate(d2, size=e1.shape[-2:], mode="nearest"), e1], 1))
        guidance, coarse = self.guidance_head(d1), self.depth_head(d1)
        return self.post_process_layer(guidance, coarse, sparse_depth)      # unet_cspn_nyu.py:386


def masked_l1(pred, target):                                                # criteria.py:27-39
    valid = (target > 0).detach()
    return (target - pred)[valid].abs().mean()


def synthetic_batch(B, H, W, device, gen):
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=device), torch.linspace(0, 1, W, device=device), indexing="ij")
    depth = (2 + 3 * yy + torch.sin(6 * xx) + 0.5 * torch.rand(B, 1, H, W, device=device, generator=gen)).clamp_min(0.1)
    rgb = torch.cat([depth / 6, yy.expand(B, 1, H, W), xx.expand(B, 1, H, W)], 1) + 0.05 * torch.randn(B, 3, H, W, device=device, generator=gen)
    sparse = depth * (torch.rand(B, 1, H, W, device=device, generator=gen) < 500.0 / (H * W))
    return torch.cat([rgb, sparse], 1), depth


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=3, help="per-GPU batch (config 5: 24 over 8 GPUs)")
    ap.add_argument("--H", type=int, default=228)
    ap.add_argument("--W", type=int, default=304)
    ap.add_argument("--lr", type=float, default=0.02)
    ap.add_argument("--model", choices=("tiny", "resnet50"), default="tiny")
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    torch.manual_seed(0)
    if args.model == "resnet50":
        if (args.H, args.W) != (228, 304):
            raise SystemExit("--model resnet50 uses the reference's 228x304 decoder pyramid (unet_cspn_nyu.py:327-332)")
        from cspn_monodepth_amd.network import unet_cspn_nyu, use_tuned_conv_db
        use_tuned_conv_db(local)            # where main.py:37 sets cudnn.benchmark: MIOpen reads the shipped result of that search
        model = unet_cspn_nyu.resnet50(reference_state_dict=False).to(device)
    else:
        model = TinyDepthNet().to(device)
    if world > 1:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)               # replaces In-Place ABN sync
        model = nn.parallel.DistributedDataParallel(model, device_ids=[local])
    opt = torch.optim.SGD(model.parameters(), lr=args.lr, momentum=0.9, weight_decay=1e-4)   # main.py:72-74
    gen = torch.Generator(device=device).manual_seed(100 + rank)
    losses = []
    t0 = None
    for it in range(args.steps):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        x, target = synthetic_batch(args.batch, args.H, args.W, device, gen)
        pred = model(x)
        loss = masked_l1(pred, target)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(1, args.steps - 5) if t0 else float("nan")
    if rank == 0:
        print("loss %.4f -> %.4f over %d steps; %.2f ms/step (batch %d/GPU x %d GPU)" % (
            losses[0], sum(losses[-5:]) / 5, args.steps, dt * 1e3, args.batch, world))
    if world > 1:
        dist.destroy_process_group()
    return losses


if __name__ == "__main__":
    main()
