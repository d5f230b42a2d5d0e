"""CPU: known-answer tests of the DCNv2 oracle (parity unpinned -> anchored on KATs and on
two independent restatements agreeing).  SURVEY.md section 8c KAT-1..4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn_v2 as odcn
from oracle import dcn_v2_c


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float64).float() * scale


def test_kat1_zero_offset_is_conv2d():
    x, w, b = _rand(2, 5, 9, 11, seed=1), _rand(7, 5, 3, 3, seed=2), _rand(7, seed=3)
    off = torch.zeros(2, 18, 9, 11)
    mask = torch.ones(2, 9, 9, 11)
    y = odcn.dcn_v2_conv(x, off, mask, w, b)
    np.testing.assert_allclose(y.numpy(), F.conv2d(x, w, b, padding=1).numpy(), atol=1e-5)
    yc = dcn_v2_c.dcn_v2_conv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), b.numpy())
    np.testing.assert_allclose(yc, y.numpy(), atol=1e-5)


@pytest.mark.parametrize('dy,dx', [(1, 0), (0, -2), (-1, 3), (2, 2)])
def test_kat2_integer_offset_is_shifted_conv(dy, dx):
    x, w = _rand(1, 3, 8, 10, seed=4), _rand(4, 3, 3, 3, seed=5)
    off = torch.zeros(1, 18, 8, 10)
    off[:, 0::2] = dy
    off[:, 1::2] = dx
    mask = torch.ones(1, 9, 8, 10)
    y = odcn.dcn_v2_conv(x, off, mask, w, None)
    p = 4
    xp = F.pad(x, (p, p, p, p))
    full = F.conv2d(xp, w)                      # 'valid' conv over the zero-extended plane
    ref = full[:, :, p - 1 + dy:p - 1 + dy + 8, p - 1 + dx:p - 1 + dx + 10]
    np.testing.assert_allclose(y.numpy(), ref.numpy(), atol=1e-5)


def test_kat3_zero_mask_gives_bias():
    x, w, b = _rand(1, 4, 6, 6, seed=6), _rand(3, 4, 3, 3, seed=7), _rand(3, seed=8)
    y = odcn.dcn_v2_conv(x, _rand(1, 18, 6, 6, seed=9), torch.zeros(1, 9, 6, 6), w, b)
    np.testing.assert_allclose(y.numpy(), b.view(1, 3, 1, 1).expand(1, 3, 6, 6).numpy(), atol=1e-6)


def test_kat4_out_of_range_and_partial_taps():
    x = torch.ones(1, 1, 4, 4)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0                         # centre tap only (k = 4)
    mask = torch.ones(1, 9, 4, 4)
    off = torch.zeros(1, 18, 4, 4)
    off[0, 8, 0, 0] = -1.0                      # y = -1  -> exactly outside -> 0
    off[0, 8, 0, 1] = -0.25                     # y = -0.25 -> 0.75 * in[0] + 0.25 * 0
    off[0, 9, 0, 2] = 1.5                       # x = 3.5 -> 0.5 * in[3] + 0.5 * 0 (corner outside)
    off[0, 9, 0, 3] = 1.0                       # x = 4.0 = W -> outside -> 0
    y = odcn.dcn_v2_conv(x, off, mask, w, None)
    np.testing.assert_allclose(y[0, 0, 0].numpy(), [0.0, 0.75, 0.5, 0.0], atol=1e-6)
    yc = dcn_v2_c.dcn_v2_conv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), None)
    np.testing.assert_allclose(yc[0, 0, 0], [0.0, 0.75, 0.5, 0.0], atol=1e-6)


@pytest.mark.parametrize('scale', [0.5, 3.0])
def test_two_restatements_agree_random(scale):
    x, w, b = _rand(2, 6, 10, 13, seed=10), _rand(5, 6, 3, 3, seed=11), _rand(5, seed=12)
    off = _rand(2, 18, 10, 13, seed=13, scale=scale)
    mask = torch.sigmoid(_rand(2, 9, 10, 13, seed=14))
    y = odcn.dcn_v2_conv(x, off, mask, w, b)
    yc = dcn_v2_c.dcn_v2_conv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), b.numpy())
    np.testing.assert_allclose(yc, y.numpy(), rtol=1e-4, atol=2e-5)
    y64 = odcn.dcn_v2_conv(x.double(), off.double(), mask.double(), w.double(), b.double())
    np.testing.assert_allclose(y.numpy(), y64.numpy(), rtol=1e-4, atol=2e-5)


def test_dcn_module_matches_functional():
    m = odcn.DCN(4, 6, (3, 3), 1, 1)
    m.conv_offset_mask.weight.data.normal_(0, 0.1)
    x = _rand(1, 4, 7, 7, seed=15)
    with torch.no_grad():
        y = m(x)
        om = F.conv2d(x, m.conv_offset_mask.weight, m.conv_offset_mask.bias, padding=1)
        y2 = odcn.dcn_v2_conv(x, om[:, :18], torch.sigmoid(om[:, 18:]), m.weight, m.bias)
    np.testing.assert_allclose(y.numpy(), y2.numpy(), atol=1e-6)


# ---- third formulation: F.grid_sample (PyTorch-maintained bilinear sampler) + F.conv2d ---------------------------
@pytest.mark.parametrize('scale,H,W,Ci,Co,seed', [(0.3, 9, 11, 5, 7, 0), (3.0, 12, 10, 4, 6, 1), (8.0, 7, 16, 3, 5, 2),
                                                  (0.0, 6, 6, 2, 3, 3), (1.5, 2, 9, 3, 4, 4), (20.0, 16, 16, 8, 8, 5)])
def test_gridsample_formulation_agrees_with_both_restatements(scale, H, W, Ci, Co, seed):
    """random offsets from sub-pixel to far outside the image (scale 20 on a 16x16 map: most taps are out of range,
    many straddle the (-1, 0) and (H-1, H) border strips): the gather-based torch oracle, the scalar-C im2col oracle
    and the grid_sample formulation give the same numbers (fp64: 1e-9; the C oracle is fp32: 2e-5)."""
    from oracle import dcn_v2_gridsample as gs
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn((2, Ci, H, W), generator=g, dtype=torch.float64)
    w = torch.randn((Co, Ci, 3, 3), generator=g, dtype=torch.float64)
    b = torch.randn((Co,), generator=g, dtype=torch.float64)
    off = torch.randn((2, 18, H, W), generator=g, dtype=torch.float64) * scale
    mask = torch.rand((2, 9, H, W), generator=g, dtype=torch.float64)
    if seed == 2:                                  # exact border hits: -1, 0, H-1, H and half-integers
        off[:, :, 0, :] = torch.round(off[:, :, 0, :] * 2) / 2
    y1 = odcn.dcn_v2_conv(x, off, mask, w, b)
    y3 = gs.dcn_v2_conv(x, off, mask, w, b)
    np.testing.assert_allclose(y3.numpy(), y1.numpy(), rtol=0, atol=1e-9)
    yc = dcn_v2_c.dcn_v2_conv(x.float().numpy(), off.float().numpy(), mask.float().numpy(), w.float().numpy(),
                              b.float().numpy())
    np.testing.assert_allclose(yc, y3.numpy(), rtol=0, atol=2e-5 * max(1.0, float(y3.abs().max())))


def test_gridsample_formulation_passes_the_kats_and_the_module_forward():
    from oracle import dcn_v2_gridsample as gs
    x = _rand(1, 4, 8, 10, seed=7).double()
    w, b = _rand(5, 4, 3, 3, seed=8).double(), _rand(5, seed=9).double()
    zero, one = torch.zeros(1, 18, 8, 10, dtype=torch.float64), torch.ones(1, 9, 8, 10, dtype=torch.float64)
    np.testing.assert_allclose(gs.dcn_v2_conv(x, zero, one, w, b).numpy(), F.conv2d(x, w, b, padding=1).numpy(),
                               atol=1e-12)                                            # KAT-1
    np.testing.assert_allclose(gs.dcn_v2_conv(x, zero, 0 * one, w, b).numpy(),
                               b.view(1, 5, 1, 1).expand(1, 5, 8, 10).numpy(), atol=1e-12)   # KAT-3
    far = zero.clone()
    far[:, 0::2] = -9.0                                                                 # every tap at y <= -1
    np.testing.assert_allclose(gs.dcn_v2_conv(x, far, one, w, None).numpy(), 0.0, atol=1e-12)  # KAT-4
    w_off, b_off = _rand(27, 4, 3, 3, seed=10).double() * 0.3, _rand(27, seed=11).double()
    np.testing.assert_allclose(gs.dcn_forward(x, w, b, w_off, b_off).numpy(),
                               odcn.dcn_forward(x, w, b, w_off, b_off).numpy(), atol=1e-9)


def test_restatements_match_the_upstream_build():
    """The pin of SURVEY row a6: the reference's OWN compiled DCNv2 CPU forward (oracle/_ref/libdcn_v2_ref.so, built by
    `make -C oracle ref` from upstream's src/cpu sources behind oracle/ref_shim.cpp) against the three restatements,
    on offsets from sub-pixel to far outside the image and exact border hits.  Skipped while the un-vendored submodule
    is absent -- DCNv2 parity then stays 'unpinned'."""
    from oracle import dcn_v2_upstream as up
    if not up.available():
        pytest.skip('oracle/_ref/libdcn_v2_ref.so not built: upstream DCNv2 sources are absent from /root/reference')
    for seed, (B, Ci, Co, H, W), scale in ((31, (2, 8, 6, 9, 11), 0.7), (32, (1, 16, 27, 12, 10), 4.0), (33, (1, 4, 4, 5, 7), 15.0)):
        x, w, b = _rand(B, Ci, H, W, seed=seed), _rand(Co, Ci, 3, 3, seed=seed + 100), _rand(Co, seed=seed + 200)
        off = _rand(B, 18, H, W, seed=seed + 300, scale=scale)
        off[:, :, 0, 0] = torch.round(off[:, :, 0, 0])           # exact integer hits incl. the border rows / columns
        mask = torch.sigmoid(_rand(B, 9, H, W, seed=seed + 400))
        want = up.dcn_v2_conv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), b.numpy())
        np.testing.assert_allclose(odcn.dcn_v2_conv(x, off, mask, w, b).numpy(), want, atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(dcn_v2_c.dcn_v2_conv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), b.numpy()), want,
                                   atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('scale,H,W,Ci,Co,seed', [(0.5, 9, 11, 6, 7, 0), (3.0, 12, 10, 4, 6, 1), (12.0, 8, 8, 3, 5, 2)])
def test_restatements_match_torchvision_deform_conv2d(scale, H, W, Ci, Co, seed):
    """Auto-skipping THIRD-PARTY pin of SURVEY row a6: `torchvision.ops.deform_conv2d(x, offset, w, b,
    padding=1, mask=mask)` is the public equivalent of upstream DCNv2 (same (dy, dx)-interleaved offset channels, same
    modulation) -- what `dla.py:513-518` computes through `DCN`.  torchvision is not installed in this image
    (the test skips); on any image that has the wheel it pins the three restatements to code none of them was derived from."""
    tv = pytest.importorskip('torchvision')
    from torchvision.ops import deform_conv2d
    from oracle import dcn_v2_gridsample as gs
    g = torch.Generator().manual_seed(500 + seed)
    x = torch.randn((2, Ci, H, W), generator=g)
    w = torch.randn((Co, Ci, 3, 3), generator=g)
    b = torch.randn((Co,), generator=g)
    off = torch.randn((2, 18, H, W), generator=g) * scale
    mask = torch.rand((2, 9, H, W), generator=g)
    want = deform_conv2d(x, off, w, b, padding=1, mask=mask).numpy()
    tol = 1e-5 * max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(odcn.dcn_v2_conv(x, off, mask, w, b).numpy(), want, rtol=0, atol=tol)
    np.testing.assert_allclose(gs.dcn_v2_conv(x, off, mask, w, b).numpy(), want, rtol=0, atol=tol)
    yc = dcn_v2_c.dcn_v2_conv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), b.numpy())
    np.testing.assert_allclose(yc, want, rtol=0, atol=tol)
    assert tv is not None
