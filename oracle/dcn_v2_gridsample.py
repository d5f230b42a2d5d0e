"""Oracle hardening: a THIRD formulation of the DCNv2 forward, built only from ops PyTorch itself maintains.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY STILL UNPINNED against upstream CharlesShang/DCNv2
(its source is absent from /root/reference, ``.gitmodules:10-13``): this file does not turn the a6 row green, it
removes "the builder's own reading" as the single point of failure -- ``F.grid_sample(mode='bilinear',
padding_mode='zeros', align_corners=True)`` is an independently specified and tested bilinear sampler whose rule
(every corner outside the image contributes zero, weights computed from the un-clamped coordinate) is exactly
upstream's ``dmcn_im2col_bilinear``: a tap at x in (-1, 0) blends the zero corner x0 = -1 with the pixel x1 = 0, a
tap at x <= -1 or x >= W has both corners outside and yields 0 (upstream's explicit early-out), a tap at
x in [W-1, W) keeps hx * in[W-1].

    col[b, ci*9 + k] = mask[b, k] * grid_sample(x[b, ci], (p_y - pad + i*dil + dy_k, p_x - pad + j*dil + dx_k))
    out              = conv2d(col, weight.view(Co, Ci*9, 1, 1), bias)           # the GEMM as a 1x1 conv

Run it in float64: the pixel -> normalised -> pixel coordinate round trip inside grid_sample then costs ~1e-16
relative, far below the 1e-9 the comparison with ``oracle/dcn_v2.py`` asserts.
"""
import torch
import torch.nn.functional as F


def dcn_v2_conv(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    dt = x.dtype
    ho = torch.arange(Ho, dtype=dt).view(1, Ho, 1)
    wo = torch.arange(Wo, dtype=dt).view(1, 1, Wo)
    taps = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            ys = ho * stride - padding + i * dilation + offset[:, 2 * k]
            xs = wo * stride - padding + j * dilation + offset[:, 2 * k + 1]
            # align_corners=True: normalised -1 / +1 are the centres of the first / last pixel
            gx = 2.0 * xs / max(W - 1, 1) - 1.0
            gy = 2.0 * ys / max(H - 1, 1) - 1.0
            grid = torch.stack((gx, gy), dim=-1)                       # [B,Ho,Wo,2] (x, y)
            val = F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
            taps.append(val * mask[:, k:k + 1])
    col = torch.stack(taps, dim=2).reshape(B, Ci * kh * kw, Ho, Wo)    # channel = ci*9 + k
    return F.conv2d(col, weight.reshape(Co, Ci * kh * kw, 1, 1), bias)


def dcn_forward(x, weight, bias, w_off, b_off):
    """upstream ``DCN.forward``: conv_offset_mask -> (o1, o2, mask) chunks -> sigmoid(mask) -> dcn_v2_conv"""
    out = F.conv2d(x, w_off, b_off, padding=1)
    return dcn_v2_conv(x, out[:, :18], torch.sigmoid(out[:, 18:]), weight, bias)
