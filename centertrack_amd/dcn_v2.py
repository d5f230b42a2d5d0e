"""Drop-in for the un-vendored ``DCNv2`` extension (SURVEY.md boundary B1).

The reference does ``from .DCNv2.dcn_v2 import DCN`` (src/lib/model/networks/dla.py:19,
necks/dlaup.py:17, resdcn.py:20, necks/msraup.py:20) and builds
``DCN(chi, cho, kernel_size=(3,3), stride=1, padding=1, dilation=1, deformable_groups=1)``
(dla.py:513), called as ``self.conv(x)`` on an NCHW fp32 tensor.  This module exports the
upstream names -- ``DCN``, ``DCNv2``, ``dcn_v2_conv`` -- with upstream's constructor
signatures and state-dict keys (``weight, bias, conv_offset_mask.weight,
conv_offset_mask.bias``), backed by ``ct_conv2d`` (offset/mask conv, sigmoid fused) and
``ct_dcn_v2`` (fused bilinear gather + fp32 MFMA contraction) of libcentertrack_hip.so.
Placing / aliasing this file as ``model/networks/DCNv2/dcn_v2.py`` makes the reference's
``dla.py`` run on the HIP kernel unchanged.

Inference only (the reference trains through upstream's backward; out of scope), CUDA
tensors only: there is no CPU fallback.  Supported geometry is what the hot path uses:
3x3, stride 1, padding 1, dilation 1, one deformable group, Cin a multiple of 32.
"""
import math

import torch
from torch import nn

from . import _lib, ops


def _check_geometry(kernel_size, stride, padding, dilation, deformable_groups, cin):
    ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
    st = stride if isinstance(stride, int) else stride[0]
    pd = padding if isinstance(padding, int) else padding[0]
    dl = dilation if isinstance(dilation, int) else dilation[0]
    if ks != (3, 3) or st != 1 or pd != 1 or dl != 1 or deformable_groups != 1:
        raise _lib.CTError('centertrack_amd DCN supports kernel 3x3, stride 1, padding 1, dilation 1, '
                           'deformable_groups 1 (the DLA-34 hot path, dla.py:513); got k=%s s=%s p=%s d=%s dg=%s'
                           % (ks, st, pd, dl, deformable_groups))
    if cin % 32:
        raise _lib.CTError('centertrack_amd DCN needs in_channels %% 32 == 0 (got %d)' % cin)


def _need_cuda(x):
    if not x.is_cuda:
        raise _lib.CTError('centertrack_amd DCN runs on an MI355X only (got a %s tensor); no CPU fallback' % x.device)
    if x.dtype != torch.float32:
        raise _lib.CTError('centertrack_amd DCN computes in fp32 (got %s)' % x.dtype)


def _om_view(offset, mask):
    """NCHW offset [B,18,H,W] + mask [B,9,H,W] -> the kernel's NHWC offset/mask map [B,H,W,32]"""
    B, _, H, W = offset.shape
    om = ops.new_view(B, H, W, 32, offset.device)
    lib = _lib.load()
    st = _lib.stream_ptr()
    cat = torch.cat((offset, mask), 1).contiguous()
    _lib.check(lib.ct_nchw_to_nhwc(cat.data_ptr(), B, 27, H, W, om.ptr, om.ld, st), 'ct_nchw_to_nhwc')
    return ops.View(om.buf, 0, 27)


def dcn_v2_conv(input, offset, mask, weight, bias, stride=1, padding=1, dilation=1, deformable_groups=1):
    """Upstream ``dcn_v2_conv = _DCNv2.apply`` (forward only).  offset [B,18,H,W] with (dy,dx)
    interleaved per tap, mask [B,9,H,W] (already sigmoid-ed), weight [Co,Ci,3,3], bias [Co]."""
    _need_cuda(input)
    _check_geometry(tuple(weight.shape[2:]), stride, padding, dilation, deformable_groups, input.shape[1])
    x = ops.view_from_nchw(input)
    om = _om_view(offset.float(), mask.float())
    wp = ops.pack_weight(weight.detach())
    out = ops.dcn_v2(x, om, wp, weight.shape[0], shift=None if bias is None else bias.detach().contiguous())
    return ops.view_to_nchw(out)


class DCNv2(nn.Module):
    """Upstream ``DCNv2``: ``forward(input, offset, mask)`` with caller-provided offsets."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        _check_geometry(kernel_size, stride, padding, dilation, deformable_groups, in_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()
        self._packed = {}

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        self.bias.data.zero_()

    def _pack(self, name, t):
        """MFMA-fragment packing of a weight, redone only when the parameter changes."""
        key = (t.data_ptr(), t._version, t.device)
        hit = self._packed.get(name)
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_weight(t.detach()))
            self._packed[name] = hit
        return hit[1]

    @torch.no_grad()
    def forward(self, input, offset, mask):
        _need_cuda(input)
        x = ops.view_from_nchw(input)
        om = _om_view(offset.float(), mask.float())
        out = ops.dcn_v2(x, om, self._pack('weight', self.weight), self.out_channels, shift=self.bias.detach())
        return ops.view_to_nchw(out)


class DCN(DCNv2):
    """Upstream ``DCN``: owns ``conv_offset_mask`` (Conv2d Cin -> 27, zero-initialised);
    ``forward(x)`` = offset/mask conv -> chunk(o1,o2,mask) -> sigmoid(mask) -> deformable conv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        channels_ = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(self.in_channels, channels_, kernel_size=self.kernel_size,
                                          stride=self.stride, padding=self.padding, bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    @torch.no_grad()
    def forward(self, input):
        _need_cuda(input)
        x = ops.view_from_nchw(input)
        B, H, W = x.N, x.H, x.W
        # ``o1, o2, mask = chunk(out, 3, 1); offset = cat(o1, o2)`` is out[:, :18]; the sigmoid of
        # channels 18..26 is fused into the conv epilogue
        om = ops.conv2d(x, self._pack('w_off', self.conv_offset_mask.weight), 27, 3, 1,
                        shift=self.conv_offset_mask.bias.detach(), sig=(18, 27),
                        out=ops.new_view(B, H, W, 32, input.device))
        out = ops.dcn_v2(x, ops.View(om.buf, 0, 27), self._pack('weight', self.weight), self.out_channels,
                         shift=self.bias.detach())
        return ops.view_to_nchw(out)
