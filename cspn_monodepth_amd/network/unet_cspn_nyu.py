"""Host model of BASELINE config 5: the topology of the reference's network/unet_cspn_nyu.py:295-387 (ResNet-50
encoder on a 4-channel RGB-D input, guided up-projection decoder, 1-channel coarse-depth head + bias-free 12-channel
affinity head, CSPN_new.AffinityPropagate(24, 3) on top), re-hosted on stock PyTorch-ROCm ops.

This file is the CALLER of the hot path, not the hot path: every convolution / batch-norm here is a stock
``torch.nn`` op (north_star: "the UNet encoder/decoder and affinity head run as stock PyTorch-ROCm ops").  The two
pieces of this package it uses are

* ``network.up_pooling.up_pooling`` — the zero-insertion un-pooling of the decoder stages as one HIP kernel, instead
  of nearest-upsampling times a checkerboard mask filled by an O(H*W) Python loop (unet_cspn_nyu.py:202-213, repeated
  in every decoder block :138-284), and
* ``post_process.CSPN_new.AffinityPropagate`` — the HIP CSPN forward / backward (unet_cspn_nyu.py:357-358, :386).

Parameter names and shapes equal the reference's (``state_dict`` keys of ``resnet50()`` there load here with
``strict=True``), including the decoder modules it constructs but never calls (up_proj_layer1..4, conv3,
:315-328) — pass ``reference_state_dict=False`` to leave those out (≈ 62 M parameters that would only sit in DDP's
buckets).  Multi-GPU: wrap with ``nn.SyncBatchNorm.convert_sync_batchnorm`` + ``DistributedDataParallel`` (one
process per GPU over RCCL) — that replaces the reference's thread-per-GPU DataParallel + In-Place ABN
(network/libs/base/encoding.py, network/libs/inplace_abn).
"""
import torch
import torch.nn as nn

from ..post_process import CSPN_new as post_process
from .up_pooling import up_pooling

__all__ = ["ResNet", "resnet50", "resnet18", "Bottleneck", "BasicBlock", "Gudi_UpProj_Block", "Gudi_UpProj_Block_Cat",
           "Simple_Gudi_UpConv_Block_Last_Layer", "UpProj_Block", "DECODER_SIZES_NYU"]

# (oheight, owidth) of the five decoder stages for a 228 x 304 input (unet_cspn_nyu.py:327-332)
DECODER_SIZES_NYU = ((15, 19), (29, 38), (57, 76), (114, 152), (228, 304))


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class _Residual(nn.Module):
    """Shared forward of the two encoder block types: branch(x) + shortcut(x), then ReLU."""

    def _finish(self, out, x):
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


class BasicBlock(_Residual):                                     # unet_cspn_nyu.py:55-84
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 3, stride), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3), nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        return self._finish(self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x))))), x)


class Bottleneck(_Residual):                                     # unet_cspn_nyu.py:87-124
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        return self._finish(self.bn3(self.conv3(out)), x)


class _UpBlock(nn.Module):
    """Base of the decoder blocks: un-pool by 2 to (oheight, owidth) with the HIP kernel.

    The reference's blocks up-sample to 2H x 2W, crop to (oheight, owidth) and multiply with a mask that is 1 at even
    (h, w) (unet_cspn_nyu.py:241-251): that is zero-insertion un-pooling followed by the crop."""

    def __init__(self, oheight, owidth):
        super(_UpBlock, self).__init__()
        self.oheight, self.owidth = oheight, owidth

    def _up_pooling(self, x, scale=2):
        # Loud difference: with the constructor defaults (oheight = owidth = 0) the reference's mask loops never run
        # (range(0, 0), unet_cspn_nyu.py:208-212) and the block returns an all-zero 2H x 2W map; no caller relies on that
        # (every block of the model is built with its target size, :327-332), so it raises here instead of silently
        # un-pooling to 2H x 2W.
        if not (self.oheight and self.owidth):
            raise ValueError("decoder block built without its target size (oheight / owidth = 0): the reference returns an "
                             "all-zero map in that case; pass the size of the un-pooled map")
        return up_pooling(x, scale, self.oheight, self.owidth)


class Gudi_UpProj_Block(_UpBlock):                               # unet_cspn_nyu.py:226-260
    """un-pool -> [5x5 conv, BN, ReLU, 3x3 conv, BN] + [5x5 conv, BN] shortcut -> ReLU."""

    def __init__(self, in_channels, out_channels, oheight=0, owidth=0):
        super(Gudi_UpProj_Block, self).__init__(oheight, owidth)
        self.conv1, self.bn1 = _conv(in_channels, out_channels, 5), nn.BatchNorm2d(out_channels)
        self.conv2, self.bn2 = _conv(out_channels, out_channels, 3), nn.BatchNorm2d(out_channels)
        self.sc_conv1, self.sc_bn1 = _conv(in_channels, out_channels, 5), nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self._up_pooling(x)
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(out + self.sc_bn1(self.sc_conv1(x)))


class UpProj_Block(Gudi_UpProj_Block):                           # unet_cspn_nyu.py:127-166 (constructed, never called)
    pass


class Gudi_UpProj_Block_Cat(_UpBlock):                           # unet_cspn_nyu.py:263-300
    """As Gudi_UpProj_Block, with the encoder skip concatenated after the first conv and fused by a 3x3 conv."""

    def __init__(self, in_channels, out_channels, oheight=0, owidth=0):
        super(Gudi_UpProj_Block_Cat, self).__init__(oheight, owidth)
        self.conv1, self.bn1 = _conv(in_channels, out_channels, 5), nn.BatchNorm2d(out_channels)
        self.conv1_1, self.bn1_1 = _conv(out_channels * 2, out_channels, 3), nn.BatchNorm2d(out_channels)
        self.conv2, self.bn2 = _conv(out_channels, out_channels, 3), nn.BatchNorm2d(out_channels)
        self.sc_conv1, self.sc_bn1 = _conv(in_channels, out_channels, 5), nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, side_input):
        x = self._up_pooling(x)
        out = torch.cat((self.relu(self.bn1(self.conv1(x))), side_input), 1)
        out = self.bn2(self.conv2(self.relu(self.bn1_1(self.conv1_1(out)))))
        return self.relu(out + self.sc_bn1(self.sc_conv1(x)))


class Simple_Gudi_UpConv_Block_Last_Layer(_UpBlock):             # unet_cspn_nyu.py:203-223: un-pool + one bias-free 3x3 conv
    def __init__(self, in_channels, out_channels, oheight=0, owidth=0):
        super(Simple_Gudi_UpConv_Block_Last_Layer, self).__init__(oheight, owidth)
        self.conv1 = _conv(in_channels, out_channels, 3)

    def forward(self, x, out_channels=None):
        """out_channels < conv1.out_channels: only the first `out_channels` filters are run (the parameter keeps its shape — the
        reference's state_dict — and autograd hands the filters that did not run a zero gradient, which is what they get from the
        CSPN module anyway: it reads channels 0..7 of the 12, CSPN_new.py:29-36)."""
        x = self._up_pooling(x)
        if out_channels is None or out_channels >= self.conv1.out_channels:
            return self.conv1(x)
        c = self.conv1
        return nn.functional.conv2d(x, c.weight[:out_channels], None, c.stride, c.padding, c.dilation, c.groups)


class ResNet(nn.Module):
    """unet_cspn_nyu.py:295-387.  input [B,4,H,W] (RGB + sparse depth) -> refined depth [B,1,H,W].

    decoder_sizes: the five (oheight, owidth) pairs of the decoder stages; the default is the reference's hard-coded
    228 x 304 pyramid (:327-332).  prop_time / prop_kernel: the reference hard-codes (24, 3) (:357-358).
    `return_cspn_io=True` makes forward also return (guidance, coarse, sparse) — the tensors handed to the CSPN module.
    affinity_channels (round 6, SURVEY.md §8 f1 "emit only 8 channels instead of 12"): how many of the affinity head's 12 filters
    (:332) the forward runs.  The default 8 is all the CSPN module reads (CSPN_new.py:29-36): the head's last convolution, its
    weight-gradient work and the [B,12,H,W] write shrink by a third (27 MB less per forward at B = 24), the refined depth and the
    gradients of channels 0..7 are unchanged (the 4 dead filters received exact zeros before, and receive them now), the parameter
    and the state_dict keep the reference's [12,64,3,3] shape.  12 reproduces the reference's tensor (guidance [B,12,H,W])."""

    def __init__(self, block, layers, up_proj_block=UpProj_Block, decoder_sizes=DECODER_SIZES_NYU, prop_time=24,
                 prop_kernel=3, reference_state_dict=True, cspn_plan=None, affinity_channels=8):
        super(ResNet, self).__init__()
        self.inplanes = 64
        e = block.expansion
        self.conv1_1 = nn.Conv2d(4, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.mid_channel = 256 * e
        self.conv2, self.bn2 = _conv(512 * e, 512 * e, 3), nn.BatchNorm2d(512 * e)
        if reference_state_dict:        # built by the reference, never called in its forward (:315-326): checkpoint keys only
            m = self.mid_channel
            self.up_proj_layer1 = up_proj_block(m, m // 2)
            self.up_proj_layer2 = up_proj_block(m // 2, m // 4)
            self.up_proj_layer3 = up_proj_block(m // 4, m // 8)
            self.up_proj_layer4 = up_proj_block(m // 8, m // 16)
            self.conv3 = _conv(128, 1, 3)
        self.post_process_layer = post_process.AffinityPropagate(prop_time, prop_kernel, plan=cspn_plan)
        s = decoder_sizes
        self.gud_up_proj_layer1 = Gudi_UpProj_Block(512 * e, 256 * e, *s[0])
        self.gud_up_proj_layer2 = Gudi_UpProj_Block_Cat(256 * e, 128 * e, *s[1])
        self.gud_up_proj_layer3 = Gudi_UpProj_Block_Cat(128 * e, 64 * e, *s[2])
        self.gud_up_proj_layer4 = Gudi_UpProj_Block_Cat(64 * e, 64, *s[3])
        self.gud_up_proj_layer5 = Simple_Gudi_UpConv_Block_Last_Layer(64, 1, *s[4])       # coarse depth head
        self.gud_up_proj_layer6 = Simple_Gudi_UpConv_Block_Last_Layer(64, 12, *s[4])      # affinity head (8 of 12 used)
        if not 8 <= int(affinity_channels) <= 12:
            raise ValueError("affinity_channels must be in [8, 12] (the CSPN module reads channels 0..7)")
        self.affinity_channels = int(affinity_channels)
        self.return_cspn_io = False

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, stride),
                                       nn.BatchNorm2d(planes * block.expansion))
        stack = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        stack += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*stack)

    def unused_parameters(self):
        """Parameters the forward never touches (freeze them before wrapping in DistributedDataParallel)."""
        names = ("up_proj_layer1", "up_proj_layer2", "up_proj_layer3", "up_proj_layer4", "conv3")
        return [p for n in names if hasattr(self, n) for p in getattr(self, n).parameters()]

    def features(self, x):
        """Everything below the CSPN module: (guidance [B,affinity_channels,H,W], coarse [B,1,H,W], sparse [B,1,H,W])."""
        sparse_depth = x.narrow(1, 3, 1).clone()                       # :362
        x = self.conv1_1(x)
        skip4 = x                                                      # pre-BN stem output, 64 ch at H/2 (:364)
        x = self.layer1(self.maxpool(self.relu(self.bn1(x))))
        skip3 = x                                                      # 64e ch at H/4
        x = self.layer2(x)
        skip2 = x                                                      # 128e ch at H/8
        x = self.bn2(self.conv2(self.layer4(self.layer3(x))))
        x = self.gud_up_proj_layer1(x)
        x = self.gud_up_proj_layer2(x, skip2)
        x = self.gud_up_proj_layer3(x, skip3)
        x = self.gud_up_proj_layer4(x, skip4)
        return self.gud_up_proj_layer6(x, self.affinity_channels), self.gud_up_proj_layer5(x), sparse_depth

    def forward(self, x):
        guidance, coarse, sparse_depth = self.features(x)
        out = self.post_process_layer(guidance, coarse, sparse_depth)  # :386
        if self.return_cspn_io:
            return out, (guidance, coarse, sparse_depth)
        return out


def resnet50(pretrained=False, **kwargs):
    """unet_cspn_nyu.py:404-415.  `pretrained` would read pretrained/resnet50.pth there; no checkpoint exists on this box."""
    if pretrained:
        raise RuntimeError("no pretrained checkpoint is available here; load one with model.load_state_dict(...)")
    return ResNet(Bottleneck, [3, 4, 6, 3], UpProj_Block, **kwargs)


def resnet18(pretrained=False, **kwargs):
    """unet_cspn_nyu.py:390-401.  Deviation: the reference hard-codes the decoder widths of the ResNet-50 plan (2048 / 1024 /
    512 / 256, :327-330), so its resnet18 cannot run a forward (channel mismatch at the first decoder block); here the widths
    follow the block expansion (512 e, 256 e, ...), which is the same numbers for ResNet-50 and a working model for
    ResNet-18.  Checkpoints of the reference exist for ResNet-50 only (state_dict parity: tests, golden G13)."""
    if pretrained:
        raise RuntimeError("no pretrained checkpoint is available here; load one with model.load_state_dict(...)")
    return ResNet(BasicBlock, [2, 2, 2, 2], UpProj_Block, **kwargs)
