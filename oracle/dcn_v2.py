"""Oracle: modulated deformable convolution v2 forward (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the op's
source is the un-vendored submodule CharlesShang/DCNv2 (reference
``.gitmodules:10-13``, branch master, no pinned SHA; call sites
``src/lib/model/networks/dla.py:19,513``).  This restates its published
algorithm (``dcn_v2.py::DCN.forward`` + ``dcn_v2_im2col_cuda.cu::
modulated_deformable_im2col`` / ``dmcn_im2col_bilinear``), SURVEY.md Appendix B.

Semantics (per sample b, tap k = 3*i + j):
    off  = conv_offset_mask(x)                         [B, 27, H, W]
    dy   = off[:, 2k], dx = off[:, 2k+1]               (first 18 channels)
    m    = sigmoid(off[:, 18 + k])
    y    = ho*stride - pad + i*dil + dy ;  x = wo*stride - pad + j*dil + dx
    val  = 0 if y <= -1 or x <= -1 or y >= H or x >= W else bilinear(in, y, x)
           with zero contribution from any corner outside [0,H-1]x[0,W-1]
    col[ci*9 + k, ho*Wo + wo] = val * m
    out  = weight.view(Co, Ci*9) @ col + bias
"""
import math

import torch
import torch.nn.functional as F


def dcn_v2_conv(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1,
                deformable_groups=1):
    """Functional modulated deformable conv (upstream ``dcn_v2_conv`` / ``_DCNv2``).

    x [B,Ci,H,W]; offset [B,2*kh*kw,Ho,Wo] ((dy,dx) interleaved per tap);
    mask [B,kh*kw,Ho,Wo]; weight [Co,Ci,kh,kw]; bias [Co].  dtype follows x.
    """
    assert deformable_groups == 1, 'the hot path uses deformable_groups=1 (dla.py:513)'
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    dt = x.dtype
    ho = torch.arange(Ho, dtype=dt).view(1, Ho, 1)
    wo = torch.arange(Wo, dtype=dt).view(1, 1, Wo)
    xf = x.reshape(B, Ci, H * W)
    cols = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            ys = ho * stride - padding + i * dilation + offset[:, 2 * k]      # [B,Ho,Wo]
            xs = wo * stride - padding + j * dilation + offset[:, 2 * k + 1]
            inside = (ys > -1) & (xs > -1) & (ys < H) & (xs < W)
            y0 = torch.floor(ys)
            x0 = torch.floor(xs)
            ly = ys - y0
            lx = xs - x0
            hy = 1 - ly
            hx = 1 - lx
            y0 = y0.long()
            x0 = x0.long()
            y1 = y0 + 1
            x1 = x0 + 1

            def corner(yy, xx):
                ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1) & inside
                idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).view(B, 1, Ho * Wo)
                v = torch.gather(xf, 2, idx.expand(B, Ci, Ho * Wo))
                return v * ok.view(B, 1, Ho * Wo).to(dt)

            v1 = corner(y0, x0)
            v2 = corner(y0, x1)
            v3 = corner(y1, x0)
            v4 = corner(y1, x1)
            w1 = (hy * hx).view(B, 1, -1)
            w2 = (hy * lx).view(B, 1, -1)
            w3 = (ly * hx).view(B, 1, -1)
            w4 = (ly * lx).view(B, 1, -1)
            val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
            cols.append(val * mask[:, k].reshape(B, 1, Ho * Wo))
    col = torch.stack(cols, dim=2).reshape(B, Ci * kh * kw, Ho * Wo)   # row = ci*9 + k
    out = torch.matmul(weight.reshape(Co, Ci * kh * kw).unsqueeze(0), col)
    if bias is not None:
        out = out + bias.view(1, Co, 1)
    return out.view(B, Co, Ho, Wo)


def dcn_forward(x, weight, bias, w_off, b_off, stride=1, padding=1, dilation=1):
    """``DCN.forward`` (upstream dcn_v2.py): offset/mask conv -> chunk -> sigmoid ->
    dcn_v2_conv.  ``o1, o2, mask = chunk(out, 3, 1); offset = cat(o1, o2)`` is
    exactly out[:, :18]; mask = sigmoid(out[:, 18:])."""
    out = F.conv2d(x, w_off, b_off, stride=stride, padding=padding)
    o1, o2, m = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    mask = torch.sigmoid(m)
    return dcn_v2_conv(x, offset, mask, weight, bias, stride, padding, dilation, 1)


class DCN(torch.nn.Module):
    """Module with upstream ``DCN``'s constructor signature and state-dict keys
    (``weight, bias, conv_offset_mask.weight, conv_offset_mask.bias``) so the
    reference's ``dla.py`` can be imported with this as ``model.networks.DCNv2.
    dcn_v2.DCN`` when the golden vectors are generated."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding,
                 dilation=1, deformable_groups=1):
        super().__init__()
        kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.deformable_groups = deformable_groups
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels, kh, kw))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        n = in_channels * kh * kw
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        self.conv_offset_mask = torch.nn.Conv2d(
            in_channels, deformable_groups * 3 * kh * kw, kernel_size=(kh, kw),
            stride=stride, padding=padding, bias=True)
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, x):
        return dcn_forward(x, self.weight, self.bias, self.conv_offset_mask.weight,
                           self.conv_offset_mask.bias, self.stride, self.padding,
                           self.dilation)
