"""GPU parity tests of the individual HIP ops (through the C ABI) against CPU oracles:
dense convs vs torch fp32 F.conv2d, DCNv2 vs oracle/dcn_v2 (+ KATs), stems / pool /
up-sample vs torch fp32, decode vs oracle/decode + the reference golden vectors.
Tolerance: fp32 with a different summation order -> 2e-4 abs on O(1) values (north_star
allows 1e-3); indices / classes bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ATOL = 2e-4


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale).float()


def _close(a, b, atol=ATOL, rtol=1e-4, msg=''):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), '%s: %d/%d mismatches, max err %.3e at %s (got %r want %r)' % (
        msg, bad.sum(), bad.size, err.max(), np.unravel_index(err.argmax(), err.shape),
        a.flat[err.argmax()], b.flat[err.argmax()])


def test_library_loads_on_gpu(device):
    from centertrack_amd import _lib
    lib = _lib.load()
    assert lib.ct_version() >= 100


def test_pack_weight_layout(device):
    from centertrack_amd import ops
    w = _rand(20, 32, 3, 3, seed=1)
    p = ops.pack_weight(w.to(device)).cpu()
    NT, C16 = 2, 2
    p = p.view(9, C16, NT, 4, 16, 4)
    for tap, c16, nt, g, j, e in [(0, 0, 0, 0, 0, 0), (4, 1, 1, 3, 3, 2), (8, 0, 0, 2, 15, 1), (3, 1, 1, 1, 4, 3)]:
        co, ci = nt * 16 + j, c16 * 16 + 4 * g + e
        want = w[co, ci, tap // 3, tap % 3] if co < 20 else 0.0
        assert float(p[tap, c16, nt, g, j, e]) == float(want)
    assert float(p[:, :, 1, :, 4:, :].abs().max()) == 0.0      # couts 20..31 are zero padding


CONV_CASES = [
    # N, H, W, Cin, Cout, ks, stride, relu, res, affine, split_k
    (1, 16, 16, 16, 16, 3, 1, True, False, True, 0),      # cfg0 (BN16, TH16)
    (2, 13, 37, 16, 16, 3, 1, False, True, True, 0),      # ragged tile edges
    (1, 32, 48, 16, 32, 3, 2, True, False, True, 0),      # cfg1 stride 2 (level1)
    (1, 16, 24, 64, 27, 3, 1, False, False, False, 0),    # offset conv shape (Cout 27 -> pad 32)
    (1, 16, 16, 32, 64, 3, 2, True, False, True, 0),      # cfg2 stride 2
    (1, 12, 20, 64, 64, 3, 1, True, True, True, 0),       # cfg2 nkk=2, residual
    (1, 8, 8, 128, 128, 3, 1, True, True, True, 0),       # auto split-K (few tiles)
    (1, 8, 8, 128, 128, 3, 1, True, True, True, 1),       # same, no split
    (4, 32, 32, 64, 128, 3, 1, True, False, True, 1),     # cfg3 (BN128) when tiles >= 512
    (1, 8, 12, 448, 128, 1, 1, True, False, True, 0),     # root 1x1, nkk=4
    (1, 8, 8, 32, 64, 1, 1, False, False, True, 0),       # project 1x1, nkk=2, no relu
    (1, 4, 4, 512, 512, 3, 1, True, True, True, 0),       # level5-like, heavy split-K
    (1, 6, 10, 16, 80, 3, 1, False, False, False, 3),     # Cout 80 (5 n-tiles), forced split 3... clipped to nchunks
    (1, 2, 2, 256, 512, 3, 2, True, False, True, 0),      # tiny map
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'N%d_%dx%d_%d-%d_k%ds%d_r%d_res%d_a%d_sk%d' % c)
def test_conv2d_matches_torch(device, case):
    from centertrack_amd import ops
    N, H, W, Cin, Cout, ks, stride, relu, use_res, affine, split_k = case
    x = _rand(N, Cin, H, W, seed=3)
    w = _rand(Cout, Cin, ks, ks, seed=4, scale=(Cin * ks * ks) ** -0.5)
    scale = (torch.rand(Cout, generator=torch.Generator().manual_seed(5)) + 0.5) if affine else None
    shift = _rand(Cout, seed=6) if affine else None
    y = F.conv2d(x, w, None, stride=stride, padding=ks // 2)
    if affine:
        y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = _rand(*y.shape, seed=7) if use_res else None
    if use_res:
        y = y + res
    if relu:
        y = F.relu(y)
    # input as a channel slice of a wider buffer (exercises ld != C), output likewise
    xb = torch.zeros(N, H, W, Cin + 16)
    xb[..., 16:] = x.permute(0, 2, 3, 1)
    xv = ops.View(xb.to(device), 16, Cin)
    rv = ops.view_from_nchw(res.to(device)) if use_res else None
    ob = torch.full((N, y.shape[2], y.shape[3], (Cout + 3) // 4 * 4 + 8), -7.0).to(device)
    ov = ops.View(ob, 4, Cout)
    ops.conv2d(xv, ops.pack_weight(w.to(device)), Cout, ks, stride, out=ov,
               scale=None if scale is None else scale.to(device),
               shift=None if shift is None else shift.to(device), res=rv, relu=relu, split_k=split_k)
    torch.cuda.synchronize()
    _close(ov.to_nchw(), y, msg='conv')
    assert float(ob[..., :4].min()) == -7.0 and float(ob[..., 4 + Cout:].min()) == -7.0, 'wrote outside its slice'


@pytest.mark.parametrize('algo,N,H,W,Cin,Cout,ks,stride', [(7, 1, 19, 37, 16, 16, 3, 1), (8, 2, 9, 21, 16, 16, 3, 1),
                                                         (7, 1, 16, 32, 16, 32, 3, 2), (8, 1, 8, 8, 64, 27, 3, 1),
                                                         (1, 1, 20, 20, 16, 16, 3, 1), (5, 1, 6, 10, 64, 64, 1, 1)])
def test_conv2d_forced_row_tile_shapes(device, algo, N, H, W, Cin, Cout, ks, stride):
    """every row-tiled shape the autotuner may pick (ct_conv_desc.algo 1..8) computes the same convolution"""
    from centertrack_amd import ops
    x = _rand(N, Cin, H, W, seed=70)
    w = _rand(Cout, Cin, ks, ks, seed=71, scale=(Cin * ks * ks) ** -0.5)
    shift = _rand(Cout, seed=72)
    y = F.relu(F.conv2d(x, w, shift, stride=stride, padding=ks // 2))
    out = ops.conv2d(ops.view_from_nchw(x.to(device)), ops.pack_weight(w.to(device)), Cout, ks, stride,
                     shift=shift.to(device), relu=True, split_k=1, algo=algo)
    torch.cuda.synchronize()
    _close(out.to_nchw(), y, msg='row-tiled algo %d' % algo)


KS_CASES = [
    # ks-config id, N, H, W, Cin, Cout, ks, stride, split_k
    (0, 1, 9, 21, 128, 128, 3, 1, 0),      # 32px x 32co, 4 waves; ragged edges
    (1, 2, 5, 16, 256, 96, 3, 1, 0),       # 16px x 32co; Cout not a multiple of the tile
    (2, 1, 4, 4, 512, 512, 3, 1, 0),       # 8 waves split K (level5)
    (3, 1, 16, 16, 64, 64, 3, 1, 0),       # 32px x 64co (level2-like)
    (4, 1, 8, 8, 256, 256, 3, 1, 0),       # 32px x 32co, 8 waves
    (0, 1, 8, 12, 448, 128, 1, 1, 0),      # root 1x1 (chunk loop unrolled by 3, 7 chunks)
    (2, 1, 4, 4, 1280, 512, 1, 1, 0),      # level5 root 1x1, 8 waves (10 chunks)
    (1, 1, 8, 8, 64, 128, 1, 1, 0),        # project 1x1, a single chunk
    (0, 1, 16, 16, 64, 128, 3, 2, 0),      # stride 2 (level3 entry)
    (1, 1, 8, 8, 128, 256, 3, 2, 0),
    (3, 1, 10, 18, 128, 64, 3, 2, 0),
    (1, 1, 8, 8, 256, 64, 3, 1, 2),        # K-split kernel + global split-K on top
]


@pytest.mark.parametrize('case', KS_CASES, ids=lambda c: 'ks%d_N%d_%dx%d_%d-%d_k%ds%d_sk%d' % c)
def test_conv2d_ksplit_kernel_matches_torch(device, case):
    """the K-split-in-workgroup conv kernel (forced through ct_set_tuning) == torch fp32"""
    from centertrack_amd import _lib, ops
    ksid, N, H, W, Cin, Cout, ks, stride, split_k = case
    x = F.relu(_rand(N, Cin, H, W, seed=30))
    w = _rand(Cout, Cin, ks, ks, seed=31, scale=(Cin * ks * ks) ** -0.5)
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(32)) + 0.5
    shift = _rand(Cout, seed=33)
    y = F.conv2d(x, w, None, stride=stride, padding=ks // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = _rand(*y.shape, seed=34)
    y = F.relu(y + res)
    lib = _lib.load()
    try:
        _lib.check(lib.ct_set_tuning(b'conv_ks', ksid))
        out = ops.conv2d(ops.view_from_nchw(x.to(device)), ops.pack_weight(w.to(device)), Cout, ks, stride,
                         scale=scale.to(device), shift=shift.to(device), res=ops.view_from_nchw(res.to(device)),
                         relu=True, split_k=split_k)
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.ct_set_tuning(b'conv_ks', -1))
    _close(out.to_nchw(), y, msg='ksplit conv')


@pytest.mark.parametrize('algo,N,H,W,Cin,Cout,split_k', [
    (1, 1, 32, 64, 16, 16, 1), (2, 2, 16, 32, 16, 32, 1), (3, 1, 32, 32, 32, 64, 1), (4, 1, 16, 32, 64, 160, 1),
    (5, 1, 8, 16, 64, 64, 1), (6, 1, 32, 32, 32, 64, 1), (7, 1, 32, 64, 16, 32, 1), (8, 2, 16, 16, 48, 16, 1),
    (3, 1, 16, 16, 128, 64, 2), (0, 1, 32, 32, 64, 128, 0),                                   # global split-K, auto plan
    (101, 1, 16, 16, 64, 128, 1), (102, 1, 8, 8, 128, 256, 1), (103, 1, 4, 4, 256, 512, 1), (104, 1, 12, 20, 128, 64, 1),
    (102, 2, 6, 10, 256, 96, 2),
])
def test_conv2d_stride2_pool_side_output(device, algo, N, H, W, Cin, Cout, split_k):
    """round 3: the 3x3 stride-2 conv kernels (every row-tiled shape and every K-split shape) also write the 2x2 max-pool
    of their INPUT (Tree.downsample, dla.py:207) as a side output: bit-identical to F.max_pool2d, the conv result
    unchanged, nothing written outside the pool view's channel slice"""
    from centertrack_amd import ops
    x = _rand(N, Cin, H, W, seed=90 + algo)
    w = _rand(Cout, Cin, 3, 3, seed=91, scale=(Cin * 9) ** -0.5)
    shift = _rand(Cout, seed=92)
    y = F.relu(F.conv2d(x, w, shift, stride=2, padding=1))
    pb = torch.full((N, H // 2, W // 2, Cin + 12), -7.0).to(device)
    pv = ops.View(pb, 8, Cin)
    xv = ops.view_from_nchw(x.to(device))
    out = ops.conv2d(xv, ops.pack_weight(w.to(device)), Cout, 3, 2, shift=shift.to(device), relu=True, split_k=split_k,
                     algo=algo, pool=pv)
    torch.cuda.synchronize()
    _close(out.to_nchw(), y, msg='conv with pool side output, algo %d' % algo)
    assert torch.equal(pv.to_nchw().cpu(), F.max_pool2d(x, 2, 2)), 'pooled side output (algo %d)' % algo
    assert float(pb[..., :8].min()) == -7.0 and float(pb[..., 8 + Cin:].min()) == -7.0, 'wrote outside its slice'
    plain = ops.conv2d(xv, ops.pack_weight(w.to(device)), Cout, 3, 2, shift=shift.to(device), relu=True, split_k=split_k,
                       algo=algo)
    torch.cuda.synchronize()
    assert torch.equal(plain.to_nchw(), out.to_nchw())


@pytest.mark.parametrize('algo,N,H,W,Cin,Cout', [(201, 1, 16, 32, 64, 64), (202, 2, 9, 21, 64, 48), (201, 1, 8, 16, 128, 256),
                                                 (202, 1, 5, 7, 256, 96), (201, 1, 12, 20, 64, 1280), (203, 2, 13, 21, 64, 80),
                                                 (203, 1, 8, 16, 128, 64), (204, 1, 10, 18, 192, 40), (204, 1, 16, 16, 64, 27),
                                                 (205, 1, 9, 17, 128, 96), (206, 1, 8, 8, 256, 64), (207, 2, 4, 4, 512, 48),
                                                 (206, 1, 8, 16, 64, 32), (208, 1, 12, 20, 64, 1280), (208, 2, 9, 21, 64, 80),
                                                 (209, 1, 16, 32, 64, 256), (209, 2, 5, 7, 64, 176), (209, 1, 8, 16, 64, 48),
                                                 (210, 1, 12, 20, 64, 1280), (210, 2, 7, 9, 64, 200), (211, 1, 16, 16, 64, 304)])
def test_conv2d_winograd_matches_torch(device, algo, N, H, W, Cin, Cout):
    """Winograd F(2x2,3x3) conv (ct_conv2d algo 201 / 202) == torch fp32 conv + BN + residual + ReLU; ragged
    edges, several chunks, Cout not a multiple of the tile.  Tolerance 5e-4 abs on O(1) outputs (the transforms
    add a few ulps of rounding on top of the summation order; north_star allows 1e-3)."""
    from centertrack_amd import ops
    x = F.relu(_rand(N, Cin, H, W, seed=60))
    w = _rand(Cout, Cin, 3, 3, seed=61, scale=(Cin * 9) ** -0.5)
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(62)) + 0.5
    shift = _rand(Cout, seed=63)
    res = _rand(N, Cout, H, W, seed=64)
    y = F.relu(F.conv2d(x, w, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    wd = w.to(device)
    out = ops.conv2d(ops.view_from_nchw(x.to(device)), ops.pack_weight(wd), Cout, 3, 1, scale=scale.to(device),
                     shift=shift.to(device), res=ops.view_from_nchw(res.to(device)), relu=True, split_k=1, algo=algo,
                     w_wino=ops.pack_winograd(wd))
    torch.cuda.synchronize()
    _close(out.to_nchw(), y, atol=5e-4, msg='winograd conv')


@pytest.mark.parametrize('algo,N,H,W,Cin,Cout', [
    (1, 1, 32, 64, 16, 16), (2, 2, 16, 32, 16, 32), (3, 1, 32, 32, 32, 64), (4, 1, 16, 32, 64, 160), (5, 1, 8, 16, 64, 64),
    (6, 1, 34, 60, 32, 64), (7, 1, 32, 64, 16, 32), (8, 2, 16, 16, 48, 16), (3, 1, 34, 60, 64, 128),
    (101, 1, 16, 16, 64, 128), (102, 1, 8, 8, 128, 256), (103, 1, 4, 4, 256, 512), (104, 1, 12, 20, 128, 64),
    (102, 2, 6, 10, 256, 96), (101, 1, 34, 60, 64, 128), (103, 1, 34, 60, 256, 512),
])
def test_conv2d_stride2_fused_projection(device, algo, N, H, W, Cin, Cout):
    """round 4: Tree.project of the pooled input (dla.py:196-203,207,217-218: conv1x1 + BN of max_pool2d(x, 2, 2), no
    ReLU -- the residual tree1 adds) as a SECOND output of the 3x3 stride-2 launch, every row-tiled / K-split shape:
    equal to torch's pool -> conv1x1 -> affine, the main output and the pool side output unchanged
    (bit for bit), with and without the pool side output"""
    from centertrack_amd import ops
    x = _rand(N, Cin, H, W, seed=190 + algo)
    w = _rand(Cout, Cin, 3, 3, seed=191, scale=(Cin * 9) ** -0.5)
    shift = _rand(Cout, seed=192)
    wp = _rand(Cout, Cin, 1, 1, seed=193, scale=Cin ** -0.5)
    psc = torch.rand(Cout, generator=torch.Generator().manual_seed(194)) + 0.5
    psh = _rand(Cout, seed=195)
    y = F.relu(F.conv2d(x, w, shift, stride=2, padding=1))
    yp = F.conv2d(F.max_pool2d(x, 2, 2), wp) * psc.view(1, -1, 1, 1) + psh.view(1, -1, 1, 1)
    xv = ops.view_from_nchw(x.to(device))
    wpk, wppk = ops.pack_weight(w.to(device)), ops.pack_weight(wp.to(device))
    plain = ops.conv2d(xv, wpk, Cout, 3, 2, shift=shift.to(device), relu=True, split_k=1, algo=algo)
    for with_pool in (False, True):
        qb = torch.full((N, H // 2, W // 2, Cout + 12), -7.0).to(device)
        qv = ops.View(qb, 4, Cout)
        pv = ops.new_view(N, H // 2, W // 2, Cin, device) if with_pool else None
        out = ops.conv2d(xv, wpk, Cout, 3, 2, shift=shift.to(device), relu=True, split_k=1, algo=algo, pool=pv,
                         proj=(wppk, psc.to(device), psh.to(device), qv))
        torch.cuda.synchronize()
        assert torch.equal(plain.to_nchw(), out.to_nchw()), 'main output changed (algo %d)' % algo
        _close(out.to_nchw(), y, msg='conv, algo %d' % algo)
        _close(qv.to_nchw(), yp, msg='fused projection, algo %d' % algo)
        assert float(qb[..., :4].min()) == -7.0 and float(qb[..., 4 + Cout:].min()) == -7.0, 'wrote outside its slice'
        if with_pool:
            assert torch.equal(pv.to_nchw().cpu(), F.max_pool2d(x, 2, 2))


def test_conv2d_fused_projection_rejects_split_k(device):
    from centertrack_amd import _lib, ops
    x = ops.view_from_nchw(_rand(1, 128, 8, 8, seed=1).to(device))
    w = ops.pack_weight(_rand(64, 128, 3, 3, seed=2).to(device))
    wp = ops.pack_weight(_rand(64, 128, 1, 1, seed=3).to(device))
    with pytest.raises(_lib.CTError, match='split-K'):
        ops.conv2d(x, w, 64, 3, 2, split_k=2, algo=3, proj=(wp, None, None, ops.new_view(1, 4, 4, 64, device)))


def test_conv2d_nchw_output_sigmoid_dep(device):
    from centertrack_amd import ops
    N, H, W, Cin = 2, 8, 12, 256
    x = _rand(N, Cin, H, W, seed=8)
    for Cout, sig, dep in [(1, (0, 1), (0, 0)), (10, (0, 10), (0, 0)), (1, (0, 0), (0, 1)), (8, (0, 0), (0, 0)),
                           (80, (0, 80), (0, 0))]:
        w = _rand(Cout, Cin, 1, 1, seed=9, scale=Cin ** -0.5)
        b = _rand(Cout, seed=10)
        y = F.conv2d(x, w, b)
        if sig[1] > 0:
            y = torch.sigmoid(y)
        if dep[1] > 0:
            y = (1. / (torch.sigmoid(y) + 1e-6) - 1.) * 2.0
        out = torch.empty(N, Cout, H, W, device=device)
        ops.conv2d(ops.view_from_nchw(x.to(device)), ops.pack_weight(w.to(device)), Cout, 1, 1, out_nchw=out,
                   shift=b.to(device), sig=sig, dep=dep, depth_scale=2.0)
        _close(out, y, rtol=2e-4, msg='head Cout=%d' % Cout)


DCN_CASES = [
    # N, H, W, Cin, Cout, off_scale, split_k[, algo]
    (1, 8, 16, 64, 64, 0.5, 1),
    (1, 8, 16, 64, 64, 0.5, 1, 3264),     # 32-pixel tiles
    (2, 7, 19, 64, 96, 3.0, 1, 3264),     # ragged, out-of-range taps, Cout not a multiple of the tile
    (1, 6, 16, 128, 256, 1.0, 2, 32128),  # 32-pixel x 128-cout tiles + split-K
    (1, 6, 16, 128, 256, 1.0, 1, 128),
    (2, 7, 19, 64, 64, 3.0, 1),       # ragged, big offsets (out-of-range taps)
    (1, 8, 8, 128, 64, 1.0, 0),
    (1, 4, 4, 512, 256, 1.0, 0),      # split-K
    (1, 16, 16, 128, 128, 1.0, 1),
    (6, 32, 32, 64, 128, 1.0, 1),     # BN=128 config (>= 512 tiles)
    (1, 8, 8, 256, 64, 1.0, 4),
    (1, 8, 16, 64, 64, 0.5, 1, 43264),    # 64-channel steps (32 MFMAs per barrier)
    (2, 7, 19, 128, 96, 3.0, 1, 43264),   # ... ragged, out-of-range taps
    (1, 6, 16, 256, 256, 1.0, 2, 432128), # ... 128 couts + split-K (2 units of 64 channels per split)
    (1, 4, 4, 512, 256, 1.0, 0, 43264),   # ... heuristic split-K
    (3, 17, 30, 128, 64, 1.0, 4, 3264),   # one chunk per split (9 steps: odd), odd map, three images
]


@pytest.mark.parametrize('case', DCN_CASES, ids=lambda c: 'N%d_%dx%d_%d-%d_o%s_sk%d' % c[:7] + ('_a%d' % c[7] if len(c) > 7 else ''))
def test_dcn_v2_matches_oracle(device, case):
    from centertrack_amd import ops
    from oracle import dcn_v2 as odcn
    N, H, W, Cin, Cout, osc, split_k = case[:7]
    algo = case[7] if len(case) > 7 else 0
    x = _rand(N, Cin, H, W, seed=11)
    w = _rand(Cout, Cin, 3, 3, seed=12, scale=(Cin * 9) ** -0.5)
    off = _rand(N, 18, H, W, seed=13, scale=osc)
    mask = torch.sigmoid(_rand(N, 9, H, W, seed=14))
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(15)) + 0.5
    shift = _rand(Cout, seed=16)
    y = odcn.dcn_v2_conv(x, off, mask, w, None)
    y = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    om = torch.zeros(N, H, W, 32)
    om[..., :18] = off.permute(0, 2, 3, 1)
    om[..., 18:27] = mask.permute(0, 2, 3, 1)
    out = ops.dcn_v2(ops.view_from_nchw(x.to(device)), ops.View(om.to(device), 0, 27), ops.pack_weight(w.to(device)),
                     Cout, scale.to(device), shift.to(device), relu=True, split_k=split_k, algo=algo)
    _close(out.to_nchw(), y, msg='dcn')


def test_dcn_v2_kat_zero_offset_is_conv(device):
    from centertrack_amd import ops
    x, w = _rand(1, 64, 9, 17, seed=17), _rand(64, 64, 3, 3, seed=18, scale=1 / 24.)
    om = torch.zeros(1, 9, 17, 32)
    om[..., 18:27] = 1.0
    out = ops.dcn_v2(ops.view_from_nchw(x.to(device)), ops.View(om.to(device), 0, 27), ops.pack_weight(w.to(device)), 64)
    _close(out.to_nchw(), F.conv2d(x, w, padding=1), msg='KAT-1')
    om[..., 18:27] = 0.0                                  # KAT-3: mask 0 -> shift only
    sh = _rand(64, seed=19)
    out = ops.dcn_v2(ops.view_from_nchw(x.to(device)), ops.View(om.to(device), 0, 27), ops.pack_weight(w.to(device)), 64,
                     shift=sh.to(device))
    _close(out.to_nchw(), sh.view(1, 64, 1, 1).expand(1, 64, 9, 17), msg='KAT-3')


def test_offset_conv_plus_dcn_is_DCN_module(device):
    """conv_offset_mask -> sigmoid(mask) -> DCN == upstream DCN.forward (oracle.dcn_forward)"""
    from centertrack_amd import ops
    from oracle import dcn_v2 as odcn
    x = F.relu(_rand(2, 128, 12, 20, seed=20))
    w, b = _rand(64, 128, 3, 3, seed=21, scale=1 / 34.), _rand(64, seed=22)
    wo, bo = _rand(27, 128, 3, 3, seed=23, scale=0.02), _rand(27, seed=24, scale=0.2)
    y = odcn.dcn_forward(x, w, b, wo, bo)
    xv = ops.view_from_nchw(x.to(device))
    om = ops.conv2d(xv, ops.pack_weight(wo.to(device)), 27, 3, 1, shift=bo.to(device), sig=(18, 27),
                    out=ops.new_view(2, 12, 20, 32, device))
    out = ops.dcn_v2(xv, om, ops.pack_weight(w.to(device)), 64, shift=b.to(device))
    _close(out.to_nchw(), y, msg='DCN module')


@pytest.mark.parametrize('N,H,W,Cin,Cout,algo,split_k', [(1, 12, 20, 64, 64, 3264, 1), (2, 9, 21, 128, 64, 0, 2),
                                                       (1, 8, 8, 256, 256, 32128, 4), (1, 6, 6, 512, 256, 3264, 8),
                                                       (1, 12, 20, 64, 64, 43264, 1), (2, 9, 21, 128, 64, 43264, 2),
                                                       (1, 8, 8, 256, 256, 432128, 4)])
def test_dcn_with_fused_offset_conv_is_DCN_module(device, N, H, W, Cin, Cout, algo, split_k):
    """one launch: conv_offset_mask + sigmoid(mask) + deformable conv == upstream DCN.forward (oracle)"""
    from centertrack_amd import ops
    from oracle import dcn_v2 as odcn
    x = F.relu(_rand(N, Cin, H, W, seed=40))
    w, b = _rand(Cout, Cin, 3, 3, seed=41, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=42)
    wo, bo = _rand(27, Cin, 3, 3, seed=43, scale=0.6 * (Cin * 9) ** -0.5), _rand(27, seed=44, scale=0.3)
    y = odcn.dcn_forward(x, w, b, wo, bo)
    out = ops.dcn_v2(ops.view_from_nchw(x.to(device)), None, ops.pack_weight(w.to(device)), Cout, shift=b.to(device),
                     algo=algo, split_k=split_k, w_off=ops.pack_weight(wo.to(device)), b_off=bo.to(device))
    _close(out.to_nchw(), y, msg='fused DCN')


@pytest.mark.parametrize('N,H,W,Cin,Cout,algo,split_k', [(1, 12, 20, 64, 64, 3264, 1), (2, 9, 21, 128, 64, 0, 2),
                                                       (1, 8, 8, 256, 256, 32128, 4), (1, 6, 6, 512, 256, 3264, 8),
                                                       (2, 9, 21, 128, 64, 43264, 2), (1, 7, 33, 256, 128, 64, 1)])
def test_dcn_with_split_offset_conv_is_DCN_module(device, N, H, W, Cin, Cout, algo, split_k):
    """fuse_offset = 2: conv_offset_mask K-split over Cin / 64 chunks by the CT_DCN_OFFSETS launch (raw partial maps),
    summed + bias + mask sigmoid by the main launch == upstream DCN.forward (oracle), on every tile shape"""
    from centertrack_amd import ops
    from oracle import dcn_v2 as odcn
    x = F.relu(_rand(N, Cin, H, W, seed=140))
    w, b = _rand(Cout, Cin, 3, 3, seed=141, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=142)
    wo, bo = _rand(27, Cin, 3, 3, seed=143, scale=0.6 * (Cin * 9) ** -0.5), _rand(27, seed=144, scale=0.3)
    y = odcn.dcn_forward(x, w, b, wo, bo)
    out = ops.dcn_v2(ops.view_from_nchw(x.to(device)), None, ops.pack_weight(w.to(device)), Cout, shift=b.to(device),
                     algo=algo, split_k=split_k, w_off=ops.pack_weight(wo.to(device)), b_off=bo.to(device),
                     split_offsets=True)
    _close(out.to_nchw(), y, msg='DCN with K-split offset conv')


@pytest.mark.parametrize('N,H,W,Cin,Cout,algo,split_k', [(1, 12, 20, 64, 64, 3264, 1), (2, 9, 21, 128, 64, 0, 2),
                                                       (1, 8, 8, 256, 256, 32128, 4), (1, 6, 6, 512, 256, 3264, 8),
                                                       (3, 17, 30, 128, 128, 43264, 2), (1, 7, 33, 256, 128, 64, 1),
                                                       (2, 17, 30, 128, 64, 128, 2)])
def test_dcn_with_winograd_split_offset_conv_is_DCN_module(device, N, H, W, Cin, Cout, algo, split_k):
    """fuse_offset = 2 with ct_dcn_desc.w_off_winograd (round 6): the K-split conv_offset_mask of the CT_DCN_OFFSETS launch
    as Winograd F(2x2,3x3) tiles -- one workgroup per 64-pixel block and 64-channel chunk, raw partial maps -- summed + bias
    + mask sigmoid by the main launch == upstream DCN.forward (oracle); ragged maps, odd sizes, several images"""
    from centertrack_amd import ops
    from oracle import dcn_v2 as odcn
    x = F.relu(_rand(N, Cin, H, W, seed=240))
    w, b = _rand(Cout, Cin, 3, 3, seed=241, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=242)
    wo, bo = _rand(27, Cin, 3, 3, seed=243, scale=0.6 * (Cin * 9) ** -0.5), _rand(27, seed=244, scale=0.3)
    y = odcn.dcn_forward(x, w, b, wo, bo)
    wod = wo.to(device)
    out = ops.dcn_v2(ops.view_from_nchw(x.to(device)), None, ops.pack_weight(w.to(device)), Cout, shift=b.to(device),
                     algo=algo, split_k=split_k, w_off=ops.pack_weight(wod), b_off=bo.to(device),
                     split_offsets=True, w_off_wino=ops.pack_winograd(wod))
    _close(out.to_nchw(), y, msg='DCN with Winograd K-split offset conv')


@pytest.mark.parametrize('N,H,W,Cin,Cout,algo,split_k,conv_algo', [(1, 12, 20, 64, 64, 3264, 1, 202),
                                                                 (2, 9, 21, 128, 64, 43264, 2, 206),
                                                                 (1, 8, 8, 256, 256, 32128, 4, 0)])
def test_dcn_with_raw_offset_sums_from_another_launch(device, N, H, W, Cin, Cout, algo, split_k, conv_algo):
    """fuse_offset = 3: conv_offset_mask computed by a plain conv launch WITHOUT bias / sigmoid (here ct_conv2d's
    Winograd shapes, which have no such epilogue, and the direct kernel) into one [N,H,W,32] map; the DCN launch adds
    the bias and the mask sigmoid == upstream DCN.forward (oracle)"""
    import ctypes
    from centertrack_amd import _lib, ops
    from oracle import dcn_v2 as odcn
    lib = _lib.load()
    x = F.relu(_rand(N, Cin, H, W, seed=150))
    w, b = _rand(Cout, Cin, 3, 3, seed=151, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=152)
    wo, bo = _rand(27, Cin, 3, 3, seed=153, scale=0.6 * (Cin * 9) ** -0.5), _rand(27, seed=154, scale=0.3)
    y = odcn.dcn_forward(x, w, b, wo, bo)
    xv = ops.view_from_nchw(x.to(device))
    wod = wo.to(device)
    omv = ops.new_view(N, H, W, 32, device)
    ops.conv2d(xv, ops.pack_weight(wod), 27, 3, 1, out=omv, algo=conv_algo,
               w_wino=ops.pack_winograd(wod) if conv_algo else None)
    out = ops.new_view(N, H, W, Cout, device)
    wp, bod, bd = ops.pack_weight(w.to(device)), bo.to(device), b.to(device)
    d = ops.make_dcn_desc(xv, None, wp, Cout, None, bd, False, out, split_k=split_k, algo=algo, w_off=wp, b_off=bod,
                          om_partial=omv.buf, raw_offsets=True)
    assert d.fuse_offset == 3 and lib.ct_dcn_v2_offsets_bytes(ctypes.byref(d)) == N * H * W * 32 * 4
    d.w_off_packed = None                                   # not read in this mode
    need = lib.ct_dcn_v2_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(need, 4) // 4, device=device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), need
    _lib.check(lib.ct_dcn_v2(ctypes.byref(d), _lib.stream_ptr()), 'ct_dcn_v2')
    _close(out.to_nchw(), y, msg='DCN on raw offset sums')


def test_dcn_split_offsets_argument_errors(device):
    import ctypes
    from centertrack_amd import _lib, ops
    lib = _lib.load()
    xv = ops.view_from_nchw(torch.zeros(1, 64, 8, 16, device=device))
    wp, wop = ops.pack_weight(torch.zeros(64, 64, 3, 3, device=device)), ops.pack_weight(torch.zeros(27, 64, 3, 3, device=device))
    bo = torch.zeros(27, device=device)
    out = ops.new_view(1, 8, 16, 64, device)
    part = torch.empty(8 * 16 * 32, device=device)
    d = ops.make_dcn_desc(xv, None, wp, 64, None, None, False, out, split_k=1, w_off=wop, b_off=bo, om_partial=part)
    assert lib.ct_dcn_v2_offsets_bytes(ctypes.byref(d)) == 8 * 16 * 32 * 4
    d.om_partial_bytes = 16
    assert lib.ct_dcn_v2(ctypes.byref(d), _lib.stream_ptr()) == _lib.CT_ERR_WORKSPACE
    d.om_partial_bytes = part.numel() * 4
    d.fuse_offset = 4
    assert lib.ct_dcn_v2(ctypes.byref(d), _lib.stream_ptr()) == _lib.CT_ERR_ARG
    d.fuse_offset = 2
    assert lib.ct_dcn_v2_group(ctypes.byref(d), 1, 8, _lib.stream_ptr()) == _lib.CT_ERR_ARG
    assert lib.ct_dcn_v2(ctypes.byref(d), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize('f,split_k', [(2, 4), (2, 1), (4, 2)])
def test_dcn_with_fused_idaup_step(device, f, split_k):
    """proj DCN + BN + ReLU + `up(.) + skip` in one C-ABI call (fused into the split-K reduction when there is
    one) == the two separate ops, bit for bit, and == the oracle."""
    from centertrack_amd import ops
    from oracle import dcn_v2 as odcn
    N, H, W, Cin, Cout = 2, 6, 10, 128, 64
    x = F.relu(_rand(N, Cin, H, W, seed=50))
    w, b = _rand(Cout, Cin, 3, 3, seed=51, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=52)
    wo, bo = _rand(27, Cin, 3, 3, seed=53, scale=0.5 * (Cin * 9) ** -0.5), _rand(27, seed=54, scale=0.3)
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(55)) + 0.5
    wup = _rand(Cout, 1, 2 * f, 2 * f, seed=56)
    skip = _rand(N, Cout, H * f, W * f, seed=57)
    y = F.relu(odcn.dcn_forward(x, w, None, wo, bo) * scale.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
    y = F.conv_transpose2d(y, wup, None, stride=f, padding=f // 2, groups=Cout) + skip
    xv, sv = ops.view_from_nchw(x.to(device)), ops.view_from_nchw(skip.to(device))
    wp, wop = ops.pack_weight(w.to(device)), ops.pack_weight(wo.to(device))
    wt = ops.upsample_weight(wup.to(device))
    up_out = ops.new_view(N, H * f, W * f, Cout, device)
    ops.dcn_v2(xv, None, wp, Cout, scale=scale.to(device), shift=b.to(device), relu=True, split_k=split_k, algo=3264,
               w_off=wop, b_off=bo.to(device), up=(wt, f, sv, up_out))
    _close(up_out.to_nchw(), y, msg='fused IDAUp step')
    pr = ops.dcn_v2(xv, None, wp, Cout, scale=scale.to(device), shift=b.to(device), relu=True, split_k=split_k, algo=3264,
                    w_off=wop, b_off=bo.to(device))
    two = ops.upsample_add(pr, wt, f, sv)
    assert torch.equal(two.to_nchw(), up_out.to_nchw())


@pytest.mark.parametrize('galgo', [3264, 43264])
def test_dcn_group_launch_equals_single_launches(device, galgo):
    """ct_dcn_v2_group: four independent layers of different shapes -- fused offset conv + IDAUp step with split-K,
    offset/mask map read from HBM, fused offset conv without split, K-split offset conv (own launch) + IDAUp step
    -- in ONE gather/contraction launch and ONE
    finishing launch == the same layers launched one by one (bit for bit: same tiles, same split-K, same reduction
    order; for every tile shape a group can run on), also when the two phases are issued separately, and == the oracle."""
    import ctypes
    from centertrack_amd import _lib, ops
    from oracle import dcn_v2 as odcn
    lib = _lib.load()
    specs = [dict(N=1, H=6, W=10, Cin=128, Cout=64, split=2, fuse=True, f=2),
             dict(N=2, H=5, W=17, Cin=256, Cout=128, split=4, fuse=False, f=0),
             dict(N=1, H=12, W=20, Cin=64, Cout=64, split=1, fuse=True, f=0),
             dict(N=1, H=7, W=18, Cin=256, Cout=64, split=2, fuse='split', f=2)]
    NL = len(specs)
    descs, keep, want, single = [], [], [], []
    for i, sp in enumerate(specs):
        N, H, W, Cin, Cout = sp['N'], sp['H'], sp['W'], sp['Cin'], sp['Cout']
        x = F.relu(_rand(N, Cin, H, W, seed=60 + i))
        w, b = _rand(Cout, Cin, 3, 3, seed=70 + i, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=80 + i)
        wo, bo = _rand(27, Cin, 3, 3, seed=90 + i, scale=0.5 * (Cin * 9) ** -0.5), _rand(27, seed=100 + i, scale=0.3)
        scale = torch.rand(Cout, generator=torch.Generator().manual_seed(110 + i)) + 0.5
        y = F.relu(odcn.dcn_forward(x, w, None, wo, bo) * scale.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
        xv = ops.view_from_nchw(x.to(device))
        wp, wop = ops.pack_weight(w.to(device)), ops.pack_weight(wo.to(device))
        sc_d, b_d, bo_d = scale.to(device), b.to(device), bo.to(device)
        om = None
        if not sp['fuse']:
            om = ops.conv2d(xv, wop, 27, 3, 1, shift=bo_d, sig=(18, 27), out=ops.new_view(N, H, W, 32, device))
        up = up1 = None
        out = ops.new_view(N, H, W, Cout, device)
        if sp['f']:
            f = sp['f']
            wup, skip = _rand(Cout, 1, 2 * f, 2 * f, seed=120 + i), _rand(N, Cout, H * f, W * f, seed=130 + i)
            y = F.conv_transpose2d(y, wup, None, stride=f, padding=f // 2, groups=Cout) + skip
            wt, sv = ops.upsample_weight(wup.to(device)), ops.view_from_nchw(skip.to(device))
            up = (wt, f, sv, ops.new_view(N, H * f, W * f, Cout, device))
            up1 = (wt, f, sv, ops.new_view(N, H * f, W * f, Cout, device))
        want.append(y)
        fuse_kw = dict(w_off=wop, b_off=bo_d) if sp['fuse'] else {}
        part = torch.empty((Cin // 64) * N * H * W * 32, device=device) if sp['fuse'] == 'split' else None
        d = ops.make_dcn_desc(xv, om, wp, Cout, sc_d, b_d, True, out, split_k=sp['split'], algo=galgo,
                              up=up, om_partial=part, **fuse_kw)
        need = lib.ct_dcn_v2_group_workspace_bytes(ctypes.byref(d))
        assert (need > 0) == (sp['split'] > 1 or sp['f'] > 0)
        ws = torch.empty(max(need, 4) // 4, device=device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
        descs.append(d)
        keep += [xv, wp, wop, sc_d, b_d, bo_d, om, up, ws, out, part]
        # the same layer alone (legacy entry point, same tile shape and split)
        one = ops.dcn_v2(xv, om, wp, Cout, sc_d, b_d, relu=True, split_k=sp['split'], algo=galgo,
                         up=up1, split_offsets=sp['fuse'] == 'split', **fuse_kw)
        single.append(up1[3] if up1 is not None else one)
    arr = (_lib.DcnDesc * NL)(*descs)

    def results():
        torch.cuda.synchronize()
        return [(specs[i]['f'] and keep[11 * i + 7][3] or keep[11 * i + 9]).to_nchw().clone() for i in range(NL)]

    allp = _lib.CT_DCN_OFFSETS | _lib.CT_DCN_MAIN | _lib.CT_DCN_FINISH
    _lib.check(lib.ct_dcn_v2_group(arr, NL, allp, _lib.stream_ptr()), 'group')
    got = results()
    for i in range(NL):
        _close(got[i], want[i], msg='group layer %d vs oracle' % i)
        assert torch.equal(got[i], single[i].to_nchw()), 'group layer %d vs single launch' % i
    for r in got:
        r.zero_()
    for i in range(NL):
        (keep[11 * i + 7][3] if specs[i]['f'] else keep[11 * i + 9]).buf.zero_()
    keep[11 * 3 + 10].zero_()
    # phases issued separately, the offsets and the finishing launches with other groupings
    _lib.check(lib.ct_dcn_v2_group((_lib.DcnDesc * 2)(descs[3], descs[0]), 2, _lib.CT_DCN_OFFSETS, _lib.stream_ptr()), 'offsets')
    _lib.check(lib.ct_dcn_v2_group(arr, NL, _lib.CT_DCN_MAIN, _lib.stream_ptr()), 'main')
    _lib.check(lib.ct_dcn_v2_group((_lib.DcnDesc * 1)(descs[1]), 1, _lib.CT_DCN_FINISH, _lib.stream_ptr()), 'finish 1')
    _lib.check(lib.ct_dcn_v2_group((_lib.DcnDesc * 3)(descs[0], descs[2], descs[3]), 3, _lib.CT_DCN_FINISH, _lib.stream_ptr()), 'finish 0,2,3')
    again = results()
    for i in range(NL):
        assert torch.equal(again[i], single[i].to_nchw()), 'separate phases, layer %d' % i


def test_dcn_group_launch_with_winograd_offsets(device):
    """ct_dcn_v2_group with the Winograd OFFSETS launch (round 6): three independent layers -- offset/mask map read from HBM +
    IDAUp step behind split-K; K-split offset conv as Winograd tiles (w_off_winograd) with four K splits; the same unsplit
    -- in ONE OFFSETS, ONE MAIN and ONE FINISH launch == the oracle, and bit-identical to the same layers launched alone"""
    import ctypes
    from centertrack_amd import _lib, ops
    from oracle import dcn_v2 as odcn
    lib = _lib.load()
    specs = [dict(N=1, H=6, W=10, Cin=128, Cout=64, split=2, offs='map', f=2),
             dict(N=2, H=5, W=17, Cin=256, Cout=128, split=4, offs='wino', f=0),
             dict(N=1, H=12, W=20, Cin=64, Cout=64, split=1, offs='wino', f=0)]
    descs, keep, want, single, outs = [], [], [], [], []
    for i, sp in enumerate(specs):
        N, H, W, Cin, Cout = sp['N'], sp['H'], sp['W'], sp['Cin'], sp['Cout']
        x = F.relu(_rand(N, Cin, H, W, seed=360 + i))
        w, b = _rand(Cout, Cin, 3, 3, seed=370 + i, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=380 + i)
        wo, bo = _rand(27, Cin, 3, 3, seed=390 + i, scale=0.5 * (Cin * 9) ** -0.5), _rand(27, seed=400 + i, scale=0.3)
        scale = torch.rand(Cout, generator=torch.Generator().manual_seed(410 + i)) + 0.5
        y = F.relu(odcn.dcn_forward(x, w, None, wo, bo) * scale.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
        xv = ops.view_from_nchw(x.to(device))
        wod = wo.to(device)
        wp, wop, wow = ops.pack_weight(w.to(device)), ops.pack_weight(wod), ops.pack_winograd(wod)
        sc_d, b_d, bo_d = scale.to(device), b.to(device), bo.to(device)
        om = None
        if sp['offs'] == 'map':
            om = ops.conv2d(xv, wop, 27, 3, 1, shift=bo_d, sig=(18, 27), out=ops.new_view(N, H, W, 32, device))
        up = up1 = None
        out = ops.new_view(N, H, W, Cout, device)
        if sp['f']:
            f = sp['f']
            wup, skip = _rand(Cout, 1, 2 * f, 2 * f, seed=420 + i), _rand(N, Cout, H * f, W * f, seed=430 + i)
            y = F.conv_transpose2d(y, wup, None, stride=f, padding=f // 2, groups=Cout) + skip
            wt, sv = ops.upsample_weight(wup.to(device)), ops.view_from_nchw(skip.to(device))
            up = (wt, f, sv, ops.new_view(N, H * f, W * f, Cout, device))
            up1 = (wt, f, sv, ops.new_view(N, H * f, W * f, Cout, device))
        want.append(y)
        wino = sp['offs'] == 'wino'
        fuse_kw = dict(w_off=wop, b_off=bo_d) if wino else {}
        part = torch.empty((Cin // 64) * N * H * W * 32, device=device) if wino else None
        d = ops.make_dcn_desc(xv, om, wp, Cout, sc_d, b_d, True, out, split_k=sp['split'], algo=3264, up=up, om_partial=part,
                              w_off_wino=wow if wino else None, **fuse_kw)
        assert bool(d.w_off_winograd) == wino
        need = lib.ct_dcn_v2_group_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(need, 4) // 4, device=device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
        descs.append(d)
        outs.append(up[3] if up is not None else out)
        keep += [xv, wp, wop, wow, sc_d, b_d, bo_d, om, up, ws, out, part]
        one = ops.dcn_v2(xv, om, wp, Cout, sc_d, b_d, relu=True, split_k=sp['split'], algo=3264, up=up1, split_offsets=wino,
                         w_off_wino=wow if wino else None, **fuse_kw)
        single.append(up1[3] if up1 is not None else one)
    arr = (_lib.DcnDesc * len(specs))(*descs)
    _lib.check(lib.ct_dcn_v2_group(arr, len(specs), _lib.CT_DCN_OFFSETS | _lib.CT_DCN_MAIN | _lib.CT_DCN_FINISH, _lib.stream_ptr()), 'group')
    torch.cuda.synchronize()
    for i in range(len(specs)):
        got = outs[i].to_nchw()
        _close(got, want[i], msg='group layer %d vs oracle' % i)
        assert torch.equal(got, single[i].to_nchw()), 'group layer %d vs single launch' % i


@pytest.mark.parametrize('palgo,algo,slots', [(53264, 3264, 5), (53264, 3264, 23), (53264, 3264, 1024), (532128, 32128, 9),
                                              (63264, 43264, 6), (632128, 432128, 1024)])
def test_dcn_persistent_launch_is_bit_identical(device, palgo, algo, slots):
    """The persistent MAIN launch (round 6: algo 5xxxx / 6xxxx -- `dcn_slots` resident workgroups, each striding over the pixel
    tiles of one (cout block, K split) column with gathers, weight loads and the next tile's sampling table running across
    tile boundaries) == the one-workgroup-per-tile launch of the same tile shape bit for bit, and == the oracle: three layers
    of a group (offset/mask map from HBM + IDAUp step; Winograd K-split offsets + split-K; K-split offsets, unsplit), with so
    few slots that every workgroup walks through many tiles (several images, ragged maps) and with more slots than tiles"""
    import ctypes
    from centertrack_amd import _lib, ops
    from oracle import dcn_v2 as odcn
    lib = _lib.load()
    wide = algo in (43264, 432128)               # 64-channel steps: an even number of 64-channel units per split
    specs = [dict(N=2, H=9, W=21, Cin=128, Cout=64, split=1, offs='map', f=2),
             dict(N=3, H=5, W=17, Cin=256, Cout=128, split=1 if wide else 2, offs='wino', f=0),
             dict(N=2, H=12, W=40, Cin=128 if wide else 64, Cout=64, split=1, offs='split', f=0)]
    res = {}
    try:
        for which in (algo, palgo):
            assert lib.ct_set_tuning(b'dcn_slots', slots) == 0
            descs, keep, outs, want = [], [], [], []
            for i, sp in enumerate(specs):
                N, H, W, Cin, Cout = sp['N'], sp['H'], sp['W'], sp['Cin'], sp['Cout']
                x = F.relu(_rand(N, Cin, H, W, seed=560 + i))
                w, b = _rand(Cout, Cin, 3, 3, seed=570 + i, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=580 + i)
                wo, bo = _rand(27, Cin, 3, 3, seed=590 + i, scale=0.5 * (Cin * 9) ** -0.5), _rand(27, seed=600 + i, scale=0.3)
                scale = torch.rand(Cout, generator=torch.Generator().manual_seed(610 + i)) + 0.5
                y = F.relu(odcn.dcn_forward(x, w, None, wo, bo) * scale.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
                xv = ops.view_from_nchw(x.to(device))
                wod = wo.to(device)
                wp, wop, wow = ops.pack_weight(w.to(device)), ops.pack_weight(wod), ops.pack_winograd(wod)
                sc_d, b_d, bo_d = scale.to(device), b.to(device), bo.to(device)
                om = None
                if sp['offs'] == 'map':
                    om = ops.conv2d(xv, wop, 27, 3, 1, shift=bo_d, sig=(18, 27), out=ops.new_view(N, H, W, 32, device))
                up = None
                out = ops.new_view(N, H, W, Cout, device)
                if sp['f']:
                    f = sp['f']
                    wup, skip = _rand(Cout, 1, 2 * f, 2 * f, seed=620 + i), _rand(N, Cout, H * f, W * f, seed=630 + i)
                    y = F.conv_transpose2d(y, wup, None, stride=f, padding=f // 2, groups=Cout) + skip
                    up = (ops.upsample_weight(wup.to(device)), f, ops.view_from_nchw(skip.to(device)), ops.new_view(N, H * f, W * f, Cout, device))
                want.append(y)
                own = sp['offs'] != 'map'
                fuse_kw = dict(w_off=wop, b_off=bo_d) if own else {}
                part = torch.empty((Cin // 64) * N * H * W * 32, device=device) if own else None
                d = ops.make_dcn_desc(xv, om, wp, Cout, sc_d, b_d, True, out, split_k=sp['split'], algo=which, up=up, om_partial=part,
                                      w_off_wino=wow if sp['offs'] == 'wino' else None, **fuse_kw)
                need = ctypes.c_size_t(0)
                _lib.check(lib.ct_dcn_v2_group_plan(ctypes.byref(d), ctypes.byref(need), None), 'plan')
                ws = torch.empty(max(need.value, 4) // 4, device=device)
                d.workspace, d.workspace_bytes = ws.data_ptr(), need.value
                descs.append(d)
                outs.append(up[3] if up is not None else out)
                keep += [xv, wp, wop, wow, sc_d, b_d, bo_d, om, up, ws, out, part]
            arr = (_lib.DcnDesc * len(specs))(*descs)
            _lib.check(lib.ct_dcn_v2_group(arr, len(specs), _lib.CT_DCN_OFFSETS | _lib.CT_DCN_MAIN | _lib.CT_DCN_FINISH, _lib.stream_ptr()), 'group')
            torch.cuda.synchronize()
            res[which] = [o.to_nchw().clone() for o in outs]
            if which == palgo:                       # ... and alone, through ct_dcn_v2
                d1 = descs[2]
                outs[2].buf.zero_()
                _lib.check(lib.ct_dcn_v2(ctypes.byref(d1), _lib.stream_ptr()), 'single persistent launch')
                torch.cuda.synchronize()
                assert torch.equal(outs[2].to_nchw(), res[algo][2]), 'single persistent launch'
    finally:
        lib.ct_set_tuning(b'dcn_slots', 1024)
    for i in range(len(specs)):
        _close(res[palgo][i], want[i], msg='persistent layer %d vs oracle' % i)
        assert torch.equal(res[palgo][i], res[algo][i]), 'layer %d: persistent vs one workgroup per tile' % i


@pytest.mark.parametrize('N,H,W,Cin,Cout,split_k,algo', [(2, 7, 19, 64, 96, 1, 3264), (1, 4, 4, 512, 256, 8, 3264), (3, 9, 21, 256, 128, 2, 32128),
                                                       (1, 12, 40, 192, 64, 3, 3264), (2, 16, 16, 128, 64, 4, 64)])
def test_dcn_xcd_aware_order_changes_no_value(device, N, H, W, Cin, Cout, split_k, algo):
    """the XCD-aware workgroup order of the MAIN launch (round 6, knob "dcn_xcd": K splits and bands of pixel tiles per XCD, id range
    padded to a multiple of 8) decides which workgroup computes which tile and nothing else: results equal the plain order bit for
    bit -- odd split counts, fewer tiles than XCDs, several images"""
    from centertrack_amd import _lib, ops
    lib = _lib.load()
    x = ops.view_from_nchw(_rand(N, Cin, H, W, seed=711).to(device))
    w = ops.pack_weight(_rand(Cout, Cin, 3, 3, seed=712, scale=(Cin * 9) ** -0.5).to(device))
    om = torch.zeros(N, H, W, 32)
    om[..., :18] = _rand(N, H, W, 18, seed=713, scale=1.5)
    om[..., 18:27] = torch.sigmoid(_rand(N, H, W, 9, seed=714))
    omv = ops.View(om.to(device), 0, 27)
    got = {}
    try:
        for x_on in (1, 0):
            assert lib.ct_set_tuning(b'dcn_xcd', x_on) == 0
            got[x_on] = ops.dcn_v2(x, omv, w, Cout, relu=True, split_k=split_k, algo=algo).to_nchw().clone()
    finally:
        lib.ct_set_tuning(b'dcn_xcd', 1)
    assert torch.equal(got[0], got[1])
    assert got[1].abs().sum() > 0


def test_dcn_persistent_argument_errors(device):
    """shapes the persistent launch cannot run are refused before anything is launched"""
    import ctypes
    from centertrack_amd import _lib, ops
    lib = _lib.load()
    xv = ops.view_from_nchw(torch.zeros(1, 64, 8, 16, device=device))
    wp = ops.pack_weight(torch.zeros(64, 64, 3, 3, device=device))
    om = ops.new_view(1, 8, 16, 32, device)
    out = ops.new_view(1, 8, 16, 64, device)
    ws = torch.empty(1 << 16, device=device)
    for algo, split, kw in ((53264, 2, {}),                    # one 32-channel chunk per split: 9 steps
                            (63264, 1, {}),                    # one 64-channel unit
                            (53264, 1, dict(w_off=wp, b_off=torch.zeros(27, device=device)))):      # fused offset conv
        d = ops.make_dcn_desc(xv, om, wp, 64, None, None, False, out, workspace=ws, split_k=split, algo=algo, **kw)
        assert lib.ct_dcn_v2(ctypes.byref(d), _lib.stream_ptr()) == _lib.CT_ERR_ARG, (algo, split)
        assert b'persistent' in lib.ct_last_error()


@pytest.mark.parametrize('with_img,with_hm,shape', [(True, True, (2, 24, 40)), (True, False, (1, 16, 32)),
                                                   (False, False, (1, 9, 33)), (True, True, (1, 64, 96))])
def test_stem_matches_torch(device, with_img, with_hm, shape):
    from centertrack_amd import ops
    N, H, W = shape
    x, pi, ph = _rand(N, 3, H, W, seed=25), _rand(N, 3, H, W, seed=26), torch.rand(N, 1, H, W)
    ws = [_rand(16, 3, 7, 7, seed=27, scale=0.1), _rand(16, 3, 7, 7, seed=28, scale=0.1),
          _rand(16, 1, 7, 7, seed=29, scale=0.2)]
    sc = torch.rand(3, 16) + 0.5
    sh = _rand(3, 16, seed=30, scale=0.3)

    def st(inp, i):
        return F.relu(F.conv2d(inp, ws[i], padding=3) * sc[i].view(1, 16, 1, 1) + sh[i].view(1, 16, 1, 1))
    y = st(x, 0)
    if with_img:
        y = y + st(pi, 1)
    if with_hm:
        y = y + st(ph, 2)
    out = ops.stem(x.to(device), pi.to(device) if with_img else None, ph.to(device) if with_hm else None,
                   ws[0].to(device), ws[1].to(device), ws[2].to(device), sc.to(device), sh.to(device))
    _close(out.to_nchw(), y, msg='stem')


@pytest.mark.parametrize('shape', [(1, 40, 72), (2, 17, 33), (1, 64, 64)])
def test_stem_on_16_row_tiles_is_bit_identical(device, shape):
    """round 4: the stem on 16 x 32 tiles (two 8-row slabs per workgroup on one staging of the planes and weights; picked by
    ct_stem_forward when the 8-row grid would be between one and two rounds of the chip, knob `stem_rows`) computes every
    pixel with the arithmetic of the 8-row tiles: equal bit for bit, ragged heights and widths, all three stems and the two-
    launch form (x / pre_img terms, then pre_hm on top)"""
    from centertrack_amd import _lib, ops
    N, H, W = shape
    x, pi, ph = _rand(N, 3, H, W, seed=125), _rand(N, 3, H, W, seed=126), torch.rand(N, 1, H, W)
    ws = [_rand(16, 3, 7, 7, seed=127, scale=0.1), _rand(16, 3, 7, 7, seed=128, scale=0.1),
          _rand(16, 1, 7, 7, seed=129, scale=0.2)]
    sc = torch.rand(3, 16) + 0.5
    sh = _rand(3, 16, seed=130, scale=0.3)
    lib = _lib.load()
    outs = {}
    try:
        for rows in (8, 16):
            _lib.check(lib.ct_set_tuning(b'stem_rows', rows))
            o = ops.stem(x.to(device), pi.to(device), ph.to(device), ws[0].to(device), ws[1].to(device), ws[2].to(device),
                         sc.to(device), sh.to(device))
            torch.cuda.synchronize()
            outs[rows] = o.to_nchw().cpu()
    finally:
        _lib.check(lib.ct_set_tuning(b'stem_rows', 0))
    y = sum(F.relu(F.conv2d(inp, ws[i], padding=3) * sc[i].view(1, 16, 1, 1) + sh[i].view(1, 16, 1, 1))
            for i, inp in enumerate((x, pi, ph)))
    _close(outs[8], y, msg='stem, 8-row tiles')
    assert torch.equal(outs[8], outs[16]), float((outs[8] - outs[16]).abs().max())


def test_maxpool_and_upsample_add(device):
    from centertrack_amd import ops
    x = _rand(2, 32, 12, 20, seed=31)
    out = ops.maxpool2x2(ops.view_from_nchw(x.to(device)))
    assert torch.equal(out.to_nchw().cpu(), F.max_pool2d(x, 2, 2))
    for f, (h, w) in [(2, (6, 10)), (4, (3, 5)), (2, (1, 1))]:
        xs = _rand(2, 64, h, w, seed=32)
        wt = _rand(64, 1, 2 * f, 2 * f, seed=33)
        skip = _rand(2, 64, h * f, w * f, seed=34)
        y = F.conv_transpose2d(xs, wt, None, stride=f, padding=f // 2, groups=64) + skip
        o = ops.upsample_add(ops.view_from_nchw(xs.to(device)), wt.to(device), f, ops.view_from_nchw(skip.to(device)))
        _close(o.to_nchw(), y, atol=1e-5, msg='upsample f=%d' % f)


def test_layout_roundtrip(device):
    from centertrack_amd import ops
    x = _rand(2, 27, 5, 7, seed=35)
    v = ops.view_from_nchw(x.to(device))
    assert torch.equal(v.to_nchw().cpu(), x)
    assert torch.equal(ops.view_to_nchw(v).cpu(), x)


def _decode_case(device, case, maps):
    from centertrack_amd import ops
    dev = {k: v.to(device).contiguous() for k, v in maps.items()}
    dec = ops.Decoder(dev['hm'], {k: v for k, v in dev.items() if k != 'hm'}, case['K'])
    packed = dec.run()
    torch.cuda.synchronize()
    return dec, dec.unpack(packed.cpu().numpy()), dec.inds.cpu().numpy()


def test_decode_matches_oracle_and_golden(device, golden_dir):
    import scenarios as S
    from oracle import decode as odecode
    g = np.load(os.path.join(golden_dir, 'decode.npz'))
    for case in S.decode_cases():
        maps = S.make_head_maps(case)
        dec, got, inds = _decode_case(device, case, maps)
        want = odecode.generic_decode({k: v.clone() for k, v in maps.items()}, K=case['K'], return_inds=True)
        np.testing.assert_array_equal(inds, want['inds'].numpy(), err_msg=case['name'] + ' inds')
        for k, v in want.items():
            if k == 'inds':
                continue
            if k == 'kps_score':       # score * mean over 17 joints: summation order of torch.mean is not restated
                np.testing.assert_allclose(got[k], v.numpy(), rtol=1e-5, atol=1e-7, err_msg=case['name'] + '.kps_score')
                np.testing.assert_allclose(got[k], g['%s.%s' % (case['name'], k)], rtol=1e-5, atol=1e-7)
                continue
            np.testing.assert_array_equal(got[k], v.numpy(), err_msg='%s.%s' % (case['name'], k))
            np.testing.assert_array_equal(got[k], g['%s.%s' % (case['name'], k)], err_msg='golden %s.%s' % (case['name'], k))


@pytest.mark.parametrize('B,h,w,K,offset,box', [(2, 24, 40, 100, 'hp_offset', 'wh'), (3, 32, 32, 40, 'reg', 'wh'),
                                                  (1, 16, 24, 20, None, 'wh'), (2, 24, 40, 60, 'hp_offset', 'none'),
                                                  (2, 24, 40, 60, 'reg', 'wh+amodal'), (2, 24, 32, 50, None, 'ltrb'),
                                                  (2, 24, 32, 50, 'reg', 'wh+ltrb+amodal')])
def test_decode_pose_branch_variants(device, B, h, w, K, offset, box):
    """pose branch beyond the reference's batch-1 golden case: batches (the reference's expand() only works at
    batch 1; the oracle's per-image form extends it), the reg head as sub-pixel offset when there is no hp_offset
    head, no offset head at all (+0.5), other K; and every source of the gate box (decode.py:45-71): wh, ltrb
    (overrides wh), wh with an ltrb_amodal head beside it (the packed row then carries the amodal box and the gate box is
    rebuilt from the heads), no box head at all (extent of the regressed joints + 25 %).  Key points exact (they are
    selections), kps_score to 1e-5."""
    from collections import OrderedDict
    import scenarios as S
    from oracle import decode as odecode
    heads = OrderedDict([('hm', 1), ('hps', 34), ('hm_hp', 17)])
    for name, c in (('wh', 2), ('ltrb', 4), ('amodal', 4)):
        if name in box.split('+'):
            heads['ltrb_amodal' if name == 'amodal' else name] = c
    if offset == 'hp_offset':
        heads['reg'] = 2
        heads['hp_offset'] = 2
    elif offset == 'reg':
        heads['reg'] = 2
    case = dict(name='pose_var', heads=heads, B=B, h=h, w=w, K=K, seed=40 + B, hps_std=1.0 if box == 'none' else 3.0)
    maps = S.make_head_maps(case)
    if 'ltrb' in heads:                               # a proper box around the centre (random N(0,2) sides would be inside-out)
        maps['ltrb'] = maps['ltrb'].abs() * torch.tensor([-3.0, -3.0, 3.0, 3.0]).view(1, 4, 1, 1)
    maps['hm_hp'][:, 3] *= 0.15                       # a joint whose peaks are all weak: regressed joints kept
    dec, got, inds = _decode_case(device, case, maps)
    want = odecode.generic_decode({k: v.clone() for k, v in maps.items()}, K=K, return_inds=True)
    np.testing.assert_array_equal(inds, want['inds'].numpy())
    np.testing.assert_array_equal(got['hps'], want['hps'].numpy())
    if box != 'none':
        np.testing.assert_array_equal(got['bboxes'], want['bboxes'].numpy())
    np.testing.assert_allclose(got['kps_score'], want['kps_score'].numpy(), rtol=1e-5, atol=1e-7)
    kps = odecode.transpose_and_gather_feat(maps['hps'], want['inds']).view(B, K, 34).clone()
    kps[..., 0::2] += want['xs'].view(B, K, 1)
    kps[..., 1::2] += want['ys'].view(B, K, 1)
    snapped = (kps.numpy() != got['hps'])
    assert snapped.any() and not snapped[..., 6:8].any()      # joint 3 never snaps (all its peaks <= 0.2)
    others = np.delete(snapped[..., 0::2], 3, axis=-1)
    assert 0.05 < others.mean() < 0.98, 'the gate box must decide both ways (%.2f snapped)' % others.mean()


def test_decode_pose_rejects_what_it_does_not_implement(device):
    from centertrack_amd import _lib, ops
    hm = torch.rand((1, 1, 16, 16), device=device)
    hps, hm_hp = torch.randn((1, 34, 16, 16), device=device), torch.rand((1, 17, 16, 16), device=device)
    with pytest.raises(_lib.CTError):                                  # no joint heat-map
        ops.Decoder(hm, {'hps': hps, 'wh': torch.rand((1, 2, 16, 16), device=device)}, 20)


@pytest.mark.parametrize('name,C,h,w,B,K,dense', [
    ('coco_x2', 80, 128, 128, 2, 100, False),          # 128 000 slots per image: stage 2a (32 slices) + stage 2
    ('kitti_x4', 3, 96, 320, 4, 100, False),           # 1024-pixel segments, every one full: stage 2a (3 slices)
    ('nusc_x1', 10, 112, 200, 1, 100, False),          # 44 000 slots, stage 2a
    ('mot_x1', 1, 128, 128, 1, 100, False),            # ~1800 candidates: compacted and sorted in LDS by stage 2 alone
    ('dense_160', 1, 160, 160, 1, 100, True),          # 6400 peaks, 10 000 slots: stage 2a + stage 2
    ('dense_k512', 2, 160, 160, 1, 512, True),         # the same with the largest K: stage 2 gets 13 x 512 keys
    ('cluster_mot', 1, 128, 128, 1, 100, 'cluster'),   # every score inside ONE histogram bin: the general path of stage 2
    ('cluster_coco', 20, 128, 128, 2, 100, 'cluster'),  # ... and the radix-select fallback of stage 2a
])
def test_decode_selection_paths_full_size(device, name, C, h, w, B, K, dense):
    """every selection path of stage 2 / 2a (round 3: candidates in registers, linear score histogram, ranked boundary
    bin; fallbacks for clustered scores) against the oracle's generic_decode: indices, classes, scores and boxes
    bit-exact.  Scores are DISTINCT by construction (a permutation of an arithmetic sequence): exact float32 ties among
    the top K would be ordered arbitrarily by torch.topk."""
    from collections import OrderedDict
    from centertrack_amd import ops
    from oracle import decode as odecode
    g = torch.Generator().manual_seed(len(name) * 131 + C)
    n = B * C * h * w
    distinct = ((torch.randperm(n, generator=g).double() + 1) / (n + 1)).view(B, C, h, w)      # n distinct values in (0, 1)
    if dense == 'cluster':
        # ~1850 peaks per class and image (every third pixel), those of class c ALL inside histogram bin 300 + 10 c
        # (1/1024 wide), 2e-7 apart -- distinct float32 values, far more than the boundary bin's tie list holds
        hm = (distinct * 0.2).float()
        ph, pw = (h + 2) // 3, (w + 2) // 3
        for c in range(C):
            rank = torch.randperm(B * ph * pw, generator=g).double().view(B, ph, pw)
            hm[:, c, 0::3, 0::3] = ((300 + 10 * c) / 1024.0 + 1e-5 + 2e-7 * rank).float()
    elif dense:                                        # a peak on every other pixel in both directions
        hm = (distinct * 0.3).float()
        hm[:, :, 0::2, 0::2] = (0.5 + 0.45 * distinct[:, :, 0::2, 0::2]).float()
    else:
        hm = (distinct * 0.98 + 0.001).float()
    maps = OrderedDict([('hm', hm), ('reg', torch.rand((B, 2, h, w), generator=g)), ('wh', torch.rand((B, 2, h, w), generator=g) * 9),
                        ('tracking', torch.randn((B, 2, h, w), generator=g))])
    dev = {k: v.to(device).contiguous() for k, v in maps.items()}
    dec = ops.Decoder(dev['hm'], {k: v for k, v in dev.items() if k != 'hm'}, K)
    got = dec.unpack(dec.run().cpu().numpy())
    inds = dec.inds.cpu().numpy()
    want = odecode.generic_decode({k: v.clone() for k, v in maps.items()}, K=K, return_inds=True)
    np.testing.assert_array_equal(inds, want['inds'].numpy(), err_msg=name + ' inds')
    for k in ('scores', 'clses', 'xs', 'ys', 'bboxes', 'tracking'):
        np.testing.assert_array_equal(got[k], want[k].numpy(), err_msg='%s.%s' % (name, k))


@pytest.mark.parametrize('B', [1, 3])
def test_decode_writes_rows_to_pinned_host_memory_and_raises_the_flag(device, B):
    """round 3: ct_decode with host_out / done_flag -- the packed rows land in pinned host memory too (bit-identical to
    the device rows), the flag is raised once every image is done, the arrival counter is back at zero (graph replays)"""
    import scenarios as S
    from centertrack_amd import ops
    case = dict(name='host_rows', heads=S.HEAD_SETS['nusc'], B=B, h=28, w=50, K=64, seed=9)
    maps = S.make_head_maps(case)
    dev = {k: v.to(device).contiguous() for k, v in maps.items()}
    heads = {k: v for k, v in dev.items() if k != 'hm'}
    F = ops.Decoder.row_floats(heads)
    host = torch.zeros((B, 64, F), dtype=torch.float32).pin_memory()
    flag = torch.zeros((16,), dtype=torch.int32).pin_memory()
    dec = ops.Decoder(dev['hm'], heads, 64, host_out=host, done_flag=flag)
    assert dec.direct
    for rep in range(3):
        host.zero_()
        flag.zero_()
        out = dec.run()
        torch.cuda.synchronize()
        assert int(flag[0]) == 1 and int(dec.done_counter[0]) == 0
        np.testing.assert_array_equal(host.numpy(), out.cpu().numpy())
    plain = ops.Decoder(dev['hm'], heads, 64)
    np.testing.assert_array_equal(plain.run().cpu().numpy(), host.numpy())


def test_decode_nms_plateau_and_ties(device):
    """KAT-5: equal neighbours are both kept by the 3x3 NMS; exact ties order by lower
    class, then lower pixel (documented tie rule; torch leaves it unspecified)."""
    from centertrack_amd import ops
    hm = torch.full((1, 2, 12, 12), 0.01)
    hm[0, 1, 3, 3] = 0.9
    hm[0, 1, 3, 4] = 0.9                       # plateau: both survive
    hm[0, 0, 8, 8] = 0.9                       # same score in a lower class -> ranks first
    hm[0, 0, 5, 5] = 0.5
    hm[0, 0, 5, 6] = 0.4                       # suppressed by its neighbour
    dec = ops.Decoder(hm.to(device), {}, 8)
    out = dec.unpack(dec.run().cpu().numpy())
    assert out['scores'][0, :4].tolist() == pytest.approx([0.9, 0.9, 0.9, 0.5])
    assert out['clses'][0, :4].tolist() == [0, 1, 1, 0]
    assert out['xs'][0, :4].tolist() == [8, 3, 4, 5] and out['ys'][0, :4].tolist() == [8, 3, 3, 5]
    assert 0.4 not in [round(float(v), 3) for v in out['scores'][0]]


def test_decode_sparse_maps_take_the_exact_slow_path(device):
    """fewer than K strictly positive NMS survivors (near-empty heat map, exact zeros, even
    negative values): the positive prefix matches the oracle, the rest follows the documented
    order (score desc, then lower class, then lower pixel) over ALL pixels."""
    from centertrack_amd import ops
    from oracle import decode as odecode
    B, C, h, w, K = 2, 3, 20, 24, 40
    hm = torch.zeros((B, C, h, w))
    g = torch.Generator().manual_seed(77)
    for b in range(B):
        for _ in range(9):
            c, y, x = int(torch.randint(0, C, (1,), generator=g)), int(torch.randint(0, h, (1,), generator=g)), \
                int(torch.randint(0, w, (1,), generator=g))
            hm[b, c, y, x] = float(torch.rand(1, generator=g)) * 0.9 + 0.05
    hm[1, 2, 5, 5] = -0.25                        # an unsuppressed negative: ranks after every zero
    reg = torch.rand((B, 2, h, w), generator=g)
    dec = ops.Decoder(hm.to(device), {'reg': reg.to(device)}, K)
    out = dec.unpack(dec.run().cpu().numpy())
    inds = dec.inds.cpu().numpy()
    want = odecode.generic_decode({'hm': hm.clone(), 'reg': reg.clone()}, K=K, return_inds=True)
    nms = odecode.nms(hm)
    for b in range(B):
        npos = int((nms[b] > 0).sum())
        assert 0 < npos < K
        np.testing.assert_array_equal(out['scores'][b, :npos], want['scores'][b, :npos].numpy())
        np.testing.assert_array_equal(inds[b, :npos], want['inds'][b, :npos].numpy())
        np.testing.assert_array_equal(out['clses'][b, :npos], want['clses'][b, :npos].numpy())
        # the remainder: all (class, pixel) with NMS'd score +0.0 in flat order
        flat = nms[b].reshape(-1).numpy()
        zeros = np.nonzero((flat == 0) & ~np.signbit(flat))[0][:K - npos]
        got_flat = out['clses'][b, npos:].astype(np.int64) * (h * w) + inds[b, npos:]
        np.testing.assert_array_equal(got_flat, zeros)
        assert (out['scores'][b, npos:] == 0).all()


def test_decode_full_size_properties(device):
    """BASELINE-size maps (COCO 80x128x128, KITTI 3x96x320 batch 8): scores sorted, every
    index a 3x3 local maximum, k-th score == torch.topk of the NMS'd map."""
    from centertrack_amd import ops
    for (B, C, h, w) in [(2, 80, 128, 128), (8, 3, 96, 320), (1, 10, 112, 200)]:
        g = torch.Generator().manual_seed(B * C)
        hm = torch.rand((B, C, h, w), generator=g)
        dec = ops.Decoder(hm.to(device), {}, 100)
        out = dec.unpack(dec.run().cpu().numpy())
        sc = out['scores']
        assert (np.diff(sc, axis=1) <= 0).all()
        nms = hm * (F.max_pool2d(hm, 3, 1, 1) == hm).float()
        ref = torch.topk(nms.view(B, -1), 100)[0].numpy()
        np.testing.assert_array_equal(sc, ref)
        cls, ys, xs = out['clses'].astype(int), out['ys'].astype(int), out['xs'].astype(int)
        for b in range(B):
            np.testing.assert_array_equal(nms[b].numpy()[cls[b], ys[b], xs[b]], sc[b])


def test_render_pre_hm_matches_reference_golden(device, golden_dir):
    """device Gaussian splatting from the native tracker's blob list == the reference's
    _get_additional_inputs output (golden pre_hm.npz), incl. the flipped copy."""
    import ctypes
    import scenarios as S
    from centertrack_amd import _lib, fast_track as FT, ops
    g = np.load(os.path.join(golden_dir, 'pre_hm.npz'))
    lay = FT.row_layout(ops.decode_layout(['reg', 'wh', 'tracking', 'ltrb_amodal'])[0])
    ident = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    for case in S.pre_hm_cases():
        meta = case['meta']
        tracks = [t for t in case['tracks'] if t['active'] != 0]     # native tracks are always active
        ft = FT.FastTracker(-1.0, -1, 100)
        rows = np.zeros((len(tracks), 14), np.float32)
        for j, t in enumerate(sorted(tracks, key=lambda t: -t['score'])):
            rows[j, 0] = t['score']
            rows[j, 4:8] = t['bbox']
        ft.step(rows, lay, -1.0, ident)
        H, W = meta['inp_height'], meta['inp_width']
        prm = torch.zeros((1, FT.MAX_BLOBS, 3), dtype=torch.int32)
        n, _ = ft.prehm_params(case['pre_thresh'], meta['trans_input'], W, H, out=prm[0].numpy())
        cnt = torch.tensor([n], dtype=torch.int32)
        flip = case['flip_test']
        out = torch.full((2 if flip else 1, 1, H, W), -1.0, device=device)
        prm_d, cnt_d = prm.to(device), cnt.to(device)       # keep both alive across the launch
        _lib.check(_lib.load().ct_render_pre_hm(prm_d.data_ptr(), cnt_d.data_ptr(), FT.MAX_BLOBS, 1,
                                                H, W, out.data_ptr(), 1 if flip else 0, _lib.stream_ptr()))
        torch.cuda.synchronize()
        ref = g[case['name'] + '.hm']
        got = out.cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-6, case['name']
        assert (got != ref).mean() < 1e-4


# ---- device-side pre-processing ----------------------------------------------------
def _preprocess_device(img, trans, dw, dh, flip, device):
    from centertrack_amd import _lib
    from centertrack_amd.detector import MEAN, STD
    lib = _lib.load()
    lut = np.empty((3, 256), np.float32)
    mean, std = np.ascontiguousarray(MEAN.reshape(-1)), np.ascontiguousarray(STD.reshape(-1))
    _lib.check(lib.ct_preprocess_lut(mean.ctypes.data, std.ctypes.data, 3, lut.ctypes.data))
    lut_d = torch.from_numpy(lut).to(device)
    img_d = torch.from_numpy(np.ascontiguousarray(img)).to(device)
    out = torch.full((2 if flip else 1, 3, dh, dw), float('nan'), device=device)
    trans = np.ascontiguousarray(trans, np.float64)
    _lib.check(lib.ct_preprocess_device(img_d.data_ptr(), img.shape[0], img.shape[1], img.shape[1] * 3, 3,
                                        trans.ctypes.data, dw, dh, lut_d.data_ptr(), out[0].data_ptr(),
                                        out[1].data_ptr() if flip else None, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('h,w,inp_h,inp_w,flip', [(360, 480, 128, 160, False), (375, 1242, 96, 320, True),
                                                   (120, 90, 64, 64, False), (33, 47, 64, 96, True),
                                                   (1080, 1920, 544, 960, False), (37, 53, 37, 53, True)])
def test_preprocess_device_is_bit_identical_to_oracle_and_host(device, h, w, inp_h, inp_w, flip):
    """ct_preprocess_device == oracle/image.pre_process_image (numpy restatement of cv2.warpAffine + normalise)
    == ct_preprocess_image (host), bit for bit; reference crop, a rotated / sheared map (negative coordinates,
    all four border cases) and an up-scaling map."""
    import ctypes
    from centertrack_amd import _lib
    from centertrack_amd.detector import MEAN, STD
    from centertrack_amd.image import get_affine_transform, make_meta
    from oracle import image as oimage
    rs = np.random.RandomState(h + w)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    meta = make_meta(inp_h, inp_w, h, w)
    rot = get_affine_transform(np.array([w / 2., h / 2.], np.float32), max(h, w) * 0.7, 25, [inp_w, inp_h])
    zoom = get_affine_transform(np.array([w / 3., h / 2.], np.float32), max(h, w) * 0.11, -7, [inp_w, inp_h])
    for tag, t in (('crop', meta['trans_input']), ('rot', rot), ('zoom', zoom)):
        got = _preprocess_device(img, t, inp_w, inp_h, flip, device)
        want = oimage.pre_process_image(img, t, inp_w, inp_h, MEAN, STD, flip)
        np.testing.assert_array_equal(got, want, err_msg=tag)
        host = np.empty(want.shape, np.float32)            # (want may be a strided view: no empty_like)
        t64 = np.ascontiguousarray(t, np.float64)
        mean, std = np.ascontiguousarray(MEAN.reshape(-1)), np.ascontiguousarray(STD.reshape(-1))
        _lib.check(_lib.load().ct_preprocess_image(img.ctypes.data_as(ctypes.c_void_p), h, w, img.strides[0], 3,
                                                   t64.ctypes.data_as(ctypes.c_void_p), inp_w, inp_h,
                                                   mean.ctypes.data_as(ctypes.c_void_p), std.ctypes.data_as(ctypes.c_void_p),
                                                   host.ctypes.data_as(ctypes.c_void_p), int(flip)))
        np.testing.assert_array_equal(got, host, err_msg=tag + ' (host)')


def test_preprocess_device_known_answers(device):
    """identity and integer shifts copy pixels exactly; a half-pixel shift rounds to nearest (test_preprocess.py)"""
    from centertrack_amd.detector import MEAN, STD
    norm = lambda u8: ((u8 / 255. - MEAN) / STD).astype(np.float32).transpose(2, 0, 1)[None]
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    ident = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    np.testing.assert_array_equal(_preprocess_device(img, ident, 53, 37, False, device), norm(img))
    shift = np.array([[1, 0, 5], [0, 1, -3]], np.float64)
    want = np.zeros_like(img)
    want[:37 - 3, 5:] = img[3:, :53 - 5]
    np.testing.assert_array_equal(_preprocess_device(img, shift, 53, 37, False, device), norm(want))
    img2 = np.zeros((4, 6, 3), np.uint8)
    img2[:, ::2] = 10
    img2[:, 1::2] = 13
    half = np.array([[1, 0, 0.5], [0, 1, 0]], np.float64)
    want2 = np.full((4, 6, 3), 12, np.uint8)
    want2[:, 0] = 5
    np.testing.assert_array_equal(_preprocess_device(img2, half, 6, 4, False, device), norm(want2))


# ---------------------------------------------------------------------------- flip_test kernels
def _flip_merge(device, heads, B, h, w, flip_idx=None):
    """heads: list of (src [2B,C,h,w] device tensor (may be a channel slice), mode) -> list of merged [B,C,h,w]"""
    import ctypes
    from centertrack_amd import _lib
    arr = (_lib.FlipHead * len(heads))()
    outs = []
    for i, (src, mode) in enumerate(heads):
        dst = torch.full((B, src.shape[1], h, w), float('nan'), device=device)
        arr[i].src, arr[i].dst, arr[i].src_batch_stride = src.data_ptr(), dst.data_ptr(), src.stride(0)
        arr[i].C, arr[i].mode = src.shape[1], mode
        outs.append(dst)
    pairs = np.ascontiguousarray(flip_idx if flip_idx is not None else np.zeros((0, 2)), np.int32).reshape(-1, 2)
    _lib.check(_lib.load().ct_flip_merge(arr, len(heads), pairs.ctypes.data if len(pairs) else None, len(pairs),
                                         B, h, w, _lib.stream_ptr()), 'ct_flip_merge')
    torch.cuda.synchronize()
    return [o.cpu() for o in outs]


@pytest.mark.parametrize('B,h,w', [(1, 6, 10), (3, 24, 80), (2, 7, 13)])
def test_flip_merge_equals_the_reference_formulas_bitwise(device, B, h, w, golden_dir):
    """ct_flip_merge == Detector._flip_output (detector.py:311-332) per stream, bit for bit: averaged heads,
    sign-flipped amodel_offset, joint-swapped hm_hp / hps (flip_lr / flip_lr_off through the oracle's
    mirror_joints, itself pinned to the reference by golden pose_flip.npz); w % 4 != 0 takes the scalar kernel;
    a channel-slice source (batch stride > C*h*w) like the heads' combined tensor."""
    from centertrack_amd import _lib
    from oracle import detector as odet
    comb = _rand(2 * B, 1 + 2 + 2 + 17 + 34 + 3, h, w, seed=B * 100 + w)
    names = [('hm', 1, _lib.CT_FLIP_AVG), ('wh', 2, _lib.CT_FLIP_AVG), ('amodel_offset', 2, _lib.CT_FLIP_NEG_EVEN),
             ('hm_hp', 17, _lib.CT_FLIP_JOINTS), ('hps', 34, _lib.CT_FLIP_JOINT_OFFSETS), ('dim', 3, _lib.CT_FLIP_AVG)]
    cd = comb.to(device)
    heads, want, c0 = [], [], 0
    for name, c, mode in names:
        v = comb[:, c0:c0 + c]
        heads.append((cd[:, c0:c0 + c], mode))
        per = []
        for b in range(B):                                 # the reference merges one image with its mirror
            a, f = v[b:b + 1], v[B + b:B + b + 1]
            if name == 'amodel_offset':
                ft = torch.flip(f, [3]).clone()
                ft[:, 0::2] *= -1
            elif name in ('hm_hp', 'hps'):
                ft = odet.mirror_joints(f, odet.COCO_FLIP_IDX, name == 'hps')
            else:
                ft = torch.flip(f, [3])
            per.append((a + ft) / 2)
        want.append(torch.cat(per, 0))
        c0 += c
    got = _flip_merge(device, heads, B, h, w, odet.COCO_FLIP_IDX)
    for (name, _, _), g_, w_ in zip(names, got, want):
        assert torch.equal(g_, w_), name
    if (B, h, w) == (1, 6, 10):                            # the reference's own flip_lr / flip_lr_off output
        gold = np.load(os.path.join(golden_dir, 'pose_flip.npz'))
        import scenarios as S
        x = S.pose_flip_inputs()
        for name, mode in (('hm_hp', _lib.CT_FLIP_JOINTS), ('hps', _lib.CT_FLIP_JOINT_OFFSETS)):
            src = torch.cat((torch.zeros_like(x[name]), x[name]), 0).to(device)      # (0 + mirrored) / 2
            out = _flip_merge(device, [(src, mode)], 1, 6, 10, odet.COCO_FLIP_IDX)[0]
            np.testing.assert_array_equal(out.numpy() * 2, gold[name])


@pytest.mark.parametrize('W', [160, 13])
def test_flip_images_mirrors_rows(device, W):
    from centertrack_amd import _lib
    x = _rand(2, 3, 9, W, seed=W).to(device)
    y = torch.full_like(x, float('nan'))
    _lib.check(_lib.load().ct_flip_images(x.data_ptr(), y.data_ptr(), 2 * 3 * 9, W, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), torch.flip(x.cpu(), [3]))


@pytest.mark.parametrize('N,H,W,heads,sig,dep', [
    (1, 12, 24, [('hm', 1), ('reg', 2), ('wh', 2), ('tracking', 2), ('ltrb_amodal', 4)], 'hm', None),
    (2, 9, 19, [('hm', 3), ('dep', 1), ('rot', 8), ('dim', 3)], 'hm', 'dep'),        # ragged tiles, an 8-channel head
    (1, 16, 16, [('wh', 2)], None, None)])
def test_heads_fused_equals_conv_relu_conv(device, N, H, W, heads, sig, dep):
    """ct_heads_fused: conv3x3 64 -> 256 + bias + ReLU + conv1x1 256 -> c + bias (+ sigmoid / depth transform) of several
    heads in one launch == the torch fp32 reference of base_model.py:24-65 + detector.py:300-308"""
    import ctypes
    from centertrack_amd import _lib, ops
    x = F.relu(_rand(N, 64, H, W, seed=200))
    nh = len(heads)
    w0 = [_rand(256, 64, 3, 3, seed=210 + i, scale=(64 * 9) ** -0.5) for i in range(nh)]
    b0 = [_rand(256, seed=220 + i, scale=0.2) for i in range(nh)]
    w2 = [_rand(c, 256, 1, 1, seed=230 + i, scale=256 ** -0.5) for i, (_, c) in enumerate(heads)]
    b2 = [_rand(c, seed=240 + i) for i, (_, c) in enumerate(heads)]
    want = []
    for i, (name, c) in enumerate(heads):
        y = F.conv2d(F.relu(F.conv2d(x, w0[i], b0[i], padding=1)), w2[i], b2[i])
        if name == sig:
            y = torch.sigmoid(y)
        if name == dep:
            y = (1. / (torch.sigmoid(y) + 1e-6) - 1.) * 2.0
        want.append(y)
    want = torch.cat(want, 1)
    ctot = want.shape[1]
    xv = ops.view_from_nchw(x.to(device))
    w0p = ops.pack_winograd(torch.cat(w0, 0).to(device))
    b0d = torch.cat(b0, 0).to(device)
    w2d = torch.zeros((nh, 8, 256), device=device)
    b2d = torch.zeros((nh, 8), device=device)
    out = torch.full((N, ctot, H, W), float('nan'), device=device)
    hd = _lib.HeadsDesc()
    hd.x, hd.N, hd.H, hd.W, hd.Cin, hd.ldx = xv.ptr, N, H, W, 64, xv.ld
    hd.w0_winograd, hd.b0, hd.nheads = w0p.data_ptr(), b0d.data_ptr(), nh
    c0 = 0
    for i, (name, c) in enumerate(heads):
        w2d[i, :c] = w2[i].reshape(c, 256).to(device)
        b2d[i, :c] = b2[i].to(device)
        hd.cout[i], hd.coff[i] = c, c0
        if name == sig:
            hd.sig_lo, hd.sig_hi = c0, c0 + c
        if name == dep:
            hd.dep_lo, hd.dep_hi = c0, c0 + c
        c0 += c
    hd.w2, hd.b2, hd.out, hd.ctot, hd.depth_scale = w2d.data_ptr(), b2d.data_ptr(), out.data_ptr(), ctot, 2.0
    _lib.check(_lib.load().ct_heads_fused(ctypes.byref(hd), _lib.stream_ptr()), 'ct_heads_fused')
    torch.cuda.synchronize()
    _close(out, want, atol=5e-4, rtol=2e-4, msg='fused heads')       # (Winograd tolerance of this suite)
    # the workgroup ORDER is a speed knob only: tile-major (0), head-major (1) and the XCD-affine head-major default (2) -- every
    # XCD takes a contiguous eighth of the (head, tile) list, also when the count is no multiple of eight -- give the same bits
    lib = _lib.load()
    try:
        for order in (0, 1, 2):
            _lib.check(lib.ct_set_tuning(b'heads_order', order), 'heads_order')
            again = torch.full((N, ctot, H, W), float('nan'), device=device)
            hd.out = again.data_ptr()
            _lib.check(lib.ct_heads_fused(ctypes.byref(hd), _lib.stream_ptr()), 'ct_heads_fused')
            torch.cuda.synchronize()
            assert torch.equal(again, out), 'heads_order %d' % order
    finally:
        _lib.check(lib.ct_set_tuning(b'heads_order', 2), 'heads_order')


@pytest.mark.parametrize('offset,nbytes', [(0, 4096), (4, 1000), (1, 37), (3, 4099)])
def test_memset_async_fills_words_and_bytes_also_inside_a_graph(device, offset, nbytes):
    """ct_memset_async is a kernel on every path (the runtime's memset NODE wrote garbage inside captured frame
    graphs, so neither the word path nor the unaligned / odd-size path may fall back to it): eager and graph-replayed fills
    of aligned and unaligned ranges leave exactly the requested bytes changed"""
    from centertrack_amd import _lib
    lib = _lib.load()
    buf = torch.full((8192,), 0x5a, dtype=torch.uint8, device=device)
    ptr = buf.data_ptr() + 16 + offset
    _lib.check(lib.ct_memset_async(ptr, 0xa7, nbytes, _lib.stream_ptr()), 'memset')
    torch.cuda.synchronize()
    want = torch.full((8192,), 0x5a, dtype=torch.uint8)
    want[16 + offset:16 + offset + nbytes] = 0xa7
    assert torch.equal(buf.cpu(), want)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            _lib.check(lib.ct_memset_async(ptr, 0x11, nbytes, _lib.stream_ptr()), 'memset in capture')
    g.replay()
    torch.cuda.synchronize()
    want[16 + offset:16 + offset + nbytes] = 0x11
    assert torch.equal(buf.cpu(), want)
