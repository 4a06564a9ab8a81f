"""Zero-insertion un-pooling of the UNet decoders on the HIP engine (cspn_unpool2d in include/cspn_hip.h).

``MyBlock`` mirrors network/unet_ours.py:131-157 (constructor ``(oheight=0, owidth=0)``, method
``_up_pooling(x, scale)``), so the decoder blocks that subclass it (UpProj_Block, Simple_Gudi_UpConv_Block, ...,
unet_ours.py:160-250) and the five copies of ``_up_pooling`` in network/unet_cspn_nyu.py:138-284 can take it unchanged:

    y[n, c, s*h, s*w] = x[n, c, h, w], zero elsewhere, cropped to (oheight, owidth)

The reference builds this with a grouped conv_transpose2d against a freshly allocated one-hot weight
(unet_ours.py:141-145) or with nearest upsampling times a checkerboard mask filled by an O(H*W) Python double loop
(unet_cspn_nyu.py:208-212, ~17 k iterations at the last decoder stage).  Here it is one streaming
kernel that also writes the zeros, and a strided-gather backward.

Non-finite inputs behave as in the reference (the inserted positions are ``x * 0``, so NaN/inf fill their s x s block);
the backward is a plain strided gather and ignores non-finite gradients at the dropped positions.

Loud difference: ``oheight == 0`` or ``owidth == 0`` (the constructor defaults) raise — the reference returns an
empty tensor (unet_ours.py:147-148) or an all-zero one (the mask loop never runs, unet_cspn_nyu.py:209) there.
"""
import torch
import torch.nn as nn
from torch.autograd.function import Function, once_differentiable

from .. import _lib
from ..functional import _device_guard, _dt, _p, _require_device, _stream

__all__ = ["up_pooling", "MyBlock"]


def _forward(x, scale, oh, ow):
    dev = _require_device(x)
    N, C, H, W = x.shape
    xc = x.contiguous()
    out = torch.empty((N, C, oh, ow), dtype=x.dtype, device=dev)
    with _device_guard(dev):
        ok = _lib.lib().cspn_unpool2d(_p(xc), _p(out), _dt(xc), N * C, H, W, scale, oh, ow, _stream(dev))
    _lib.check(ok, "cspn_unpool2d")
    return out


class _UpPooling(Function):
    @staticmethod
    def forward(ctx, x, scale, oh, ow):
        ctx.geom = (tuple(x.shape), scale, oh, ow)
        return _forward(x, scale, oh, ow)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (N, C, H, W), scale, oh, ow = ctx.geom
        dev = _require_device(grad_out)
        go = grad_out.contiguous()
        gx = torch.empty((N, C, H, W), dtype=go.dtype, device=dev)
        with _device_guard(dev):
            ok = _lib.lib().cspn_unpool2d_backward(_p(go), _p(gx), _dt(go), N * C, H, W, scale, oh, ow, _stream(dev))
        _lib.check(ok, "cspn_unpool2d_backward")
        return gx, None, None, None


def up_pooling(x, scale, oheight, owidth):
    """[N,C,H,W] -> [N,C,oheight,owidth] (unet_ours.py:138-150)."""
    if x.dim() != 4:
        raise ValueError("up_pooling: x must be [N,C,H,W], got %s" % (tuple(x.shape),))
    scale, oheight, owidth = int(scale), int(oheight), int(owidth)
    H, W = x.shape[-2:]
    if not (1 <= oheight <= scale * H and 1 <= owidth <= scale * W):
        raise ValueError("up_pooling: output %dx%d must lie within [1, %d] x [1, %d] (scale %d of a %dx%d input); the "
                         "reference's oheight/owidth = 0 defaults give an empty or all-zero tensor and are not supported"
                         % (oheight, owidth, scale * H, scale * W, scale, H, W))
    if torch.is_grad_enabled() and x.requires_grad:
        return _UpPooling.apply(x, scale, oheight, owidth)
    return _forward(x, scale, oheight, owidth)


class MyBlock(nn.Module):
    """network/unet_ours.py:131-157: base of the decoder blocks; holds the target size of the un-pooled map."""

    def __init__(self, oheight=0, owidth=0):
        super(MyBlock, self).__init__()
        self.oheight = oheight
        self.owidth = owidth

    def _up_pooling(self, x, scale):
        return up_pooling(x, scale, self.oheight, self.owidth)

    def init_weights(self):                               # unet_ours.py:152-157
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight.data)
                if m.bias is not None:
                    nn.init.constant_(m.bias.data, 0)
