"""GPU: kernels written without a GPU at hand, not selected by any pinned plan.  Skipped unless
CENTERTRACK_EXPERIMENTAL=1 -- the first GPU call of the next round runs them; a shape that passes here and wins its A/B
moves into tests/test_hip_ops.py and the tuner's candidate list.

  * algo 41664: DCNv2 on 16-pixel x 64-cout tiles, K split over the waves (`dcn16_kernel`, csrc/dcn_mfma.hip)
  * CENTERTRACK_DCN_TILE16: the frame plan with its small MAIN launches on that shape
  * algo 53264: the 32-pixel / 64-channel-step shape with the contraction on v_mfma_f32_32x32x2_f32 (`dcn32x_kernel`),
    CENTERTRACK_DCN_M32 in the frame plan"""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from test_hip_ops import _close, _rand

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('CENTERTRACK_EXPERIMENTAL', '0') != '1',
                                 reason='experimental shapes: set CENTERTRACK_EXPERIMENTAL=1')]


@pytest.fixture(scope='module')
def device():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('N,H,W,Cin,Cout,osc,split_k', [(1, 8, 16, 64, 64, 0.5, 1), (2, 7, 19, 128, 64, 3.0, 1),
                                                        (1, 6, 16, 256, 256, 1.0, 2), (1, 4, 4, 512, 256, 1.0, 0),
                                                        (1, 5, 33, 128, 128, 1.0, 2)])
@pytest.mark.parametrize('algo', [41664, 53264])
def test_dcn16_matches_oracle(device, N, H, W, Cin, Cout, osc, split_k, algo):
    """offset/mask map read from HBM; ragged widths, out-of-range taps, several cout blocks, split-K"""
    from centertrack_amd import ops
    from oracle import dcn_v2 as odcn
    x = _rand(N, Cin, H, W, seed=11)
    w = _rand(Cout, Cin, 3, 3, seed=12, scale=(Cin * 9) ** -0.5)
    off = _rand(N, 18, H, W, seed=13, scale=osc)
    mask = torch.sigmoid(_rand(N, 9, H, W, seed=14))
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(15)) + 0.5
    shift = _rand(Cout, seed=16)
    y = F.relu(odcn.dcn_v2_conv(x, off, mask, w, None) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    om = torch.zeros(N, H, W, 32)
    om[..., :18] = off.permute(0, 2, 3, 1)
    om[..., 18:27] = mask.permute(0, 2, 3, 1)
    out = ops.dcn_v2(ops.view_from_nchw(x.to(device)), ops.View(om.to(device), 0, 27), ops.pack_weight(w.to(device)),
                     Cout, scale.to(device), shift.to(device), relu=True, split_k=split_k, algo=algo)
    _close(out.to_nchw(), y, msg='dcn %d' % algo)


@pytest.mark.parametrize('N,H,W,Cin,Cout,split_k,split_offsets', [(1, 12, 20, 64, 64, 1, False), (2, 9, 21, 128, 64, 2, False),
                                                                  (1, 8, 8, 256, 256, 4, False), (2, 9, 21, 128, 64, 2, True),
                                                                  (1, 7, 33, 256, 128, 1, True)])
@pytest.mark.parametrize('algo', [41664, 53264])
def test_dcn16_with_its_offset_conv_is_DCN_module(device, N, H, W, Cin, Cout, split_k, split_offsets, algo):
    """conv_offset_mask inside the launch (one-row K-split tile) or K-split by the OFFSETS launch == upstream DCN.forward;
    and the sampling points are those of the 32-pixel shape: outputs agree to summation order"""
    from centertrack_amd import ops
    from oracle import dcn_v2 as odcn
    x = F.relu(_rand(N, Cin, H, W, seed=40))
    w, b = _rand(Cout, Cin, 3, 3, seed=41, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=42)
    wo, bo = _rand(27, Cin, 3, 3, seed=43, scale=0.6 * (Cin * 9) ** -0.5), _rand(27, seed=44, scale=0.3)
    y = odcn.dcn_forward(x, w, b, wo, bo)
    kw = dict(shift=b.to(device), split_k=split_k, w_off=ops.pack_weight(wo.to(device)), b_off=bo.to(device),
              split_offsets=split_offsets)
    xv, wp = ops.view_from_nchw(x.to(device)), ops.pack_weight(w.to(device))
    out = ops.dcn_v2(xv, None, wp, Cout, algo=algo, **kw)
    _close(out.to_nchw(), y, msg='DCN module, algo %d' % algo)
    ref = ops.dcn_v2(xv, None, wp, Cout, algo=43264, **kw)
    _close(out.to_nchw(), ref.to_nchw().cpu(), atol=2e-5, rtol=2e-5, msg='algo %d vs 43264' % algo)


@pytest.mark.parametrize('N,H,W,Cin,Cout,algo', [(2, 9, 21, 128, 64, 43264), (1, 7, 33, 256, 128, 41664)])
def test_offsets_launch_on_one_row_tiles_is_bit_identical(device, N, H, W, Cin, Cout, algo):
    """knob `dcn_offs16`: the K-split offset/mask conv launch on 16-pixel rows writes the partial maps of the 32-pixel
    launch bit for bit (same slabs, same order per pixel), so the DCN output does not change by a bit either"""
    from centertrack_amd import _lib, ops
    lib = _lib.load()
    x = F.relu(_rand(N, Cin, H, W, seed=140))
    w, b = _rand(Cout, Cin, 3, 3, seed=141, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=142)
    wo, bo = _rand(27, Cin, 3, 3, seed=143, scale=0.6 * (Cin * 9) ** -0.5), _rand(27, seed=144, scale=0.3)
    xv, wp = ops.view_from_nchw(x.to(device)), ops.pack_weight(w.to(device))
    kw = dict(shift=b.to(device), algo=algo, split_k=1, w_off=ops.pack_weight(wo.to(device)), b_off=bo.to(device),
              split_offsets=True)
    try:
        assert lib.ct_set_tuning(b'dcn_offs16', 0) == 0
        base = ops.dcn_v2(xv, None, wp, Cout, **kw).to_nchw().clone()
        assert lib.ct_set_tuning(b'dcn_offs16', 1) == 0
        got = ops.dcn_v2(xv, None, wp, Cout, **kw).to_nchw().clone()
    finally:
        lib.ct_set_tuning(b'dcn_offs16', 0)
    assert torch.equal(got, base)


@pytest.mark.parametrize('xalgo', [41664, 53264])
@pytest.mark.parametrize('f,split_k', [(2, 2), (4, 1)])
def test_dcn16_group_with_idaup_step(device, f, split_k, xalgo):
    """two layers in one MAIN launch on 16-pixel tiles + one FINISH launch (split-K reduction, BN, ReLU, IDAUp step) ==
    the same group on 32-pixel tiles to summation order, == oracle"""
    from centertrack_amd import _lib, ops
    from oracle import dcn_v2 as odcn
    lib = _lib.load()
    outs = {}
    want = []
    for galgo in (xalgo, 43264):
        descs, keep, res = [], [], []
        for i, (N, H, W, Cin, Cout, ff, sk, fuse) in enumerate([(1, 6, 10, 128, 64, f, split_k, True),
                                                                 (2, 5, 17, 256, 128, 0, 2, False)]):
            x = F.relu(_rand(N, Cin, H, W, seed=60 + i))
            w, b = _rand(Cout, Cin, 3, 3, seed=70 + i, scale=(Cin * 9) ** -0.5), _rand(Cout, seed=80 + i)
            wo, bo = _rand(27, Cin, 3, 3, seed=90 + i, scale=0.5 * (Cin * 9) ** -0.5), _rand(27, seed=100 + i, scale=0.3)
            scale = torch.rand(Cout, generator=torch.Generator().manual_seed(110 + i)) + 0.5
            xv = ops.view_from_nchw(x.to(device))
            wp, wop = ops.pack_weight(w.to(device)), ops.pack_weight(wo.to(device))
            sc_d, b_d, bo_d = scale.to(device), b.to(device), bo.to(device)
            om = None if fuse else ops.conv2d(xv, wop, 27, 3, 1, shift=bo_d, sig=(18, 27), out=ops.new_view(N, H, W, 32, device))
            out = ops.new_view(N, H, W, Cout, device)
            up = None
            if galgo == xalgo:
                y = F.relu(odcn.dcn_forward(x, w, None, wo, bo) * scale.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
            if ff:
                wup, skip = _rand(Cout, 1, 2 * ff, 2 * ff, seed=120 + i), _rand(N, Cout, H * ff, W * ff, seed=130 + i)
                if galgo == xalgo:
                    y = F.conv_transpose2d(y, wup, None, stride=ff, padding=ff // 2, groups=Cout) + skip
                up = (ops.upsample_weight(wup.to(device)), ff, ops.view_from_nchw(skip.to(device)),
                      ops.new_view(N, H * ff, W * ff, Cout, device))
            if galgo == xalgo:
                want.append(y)
            d = ops.make_dcn_desc(xv, om, wp, Cout, sc_d, b_d, True, out, split_k=sk, algo=galgo, up=up,
                                  **(dict(w_off=wop, b_off=bo_d) if fuse else {}))
            need = ctypes.c_size_t(0)
            _lib.check(lib.ct_dcn_v2_group_plan(ctypes.byref(d), ctypes.byref(need), None), 'plan')
            ws = torch.empty(max(need.value, 4) // 4, device=device)
            d.workspace, d.workspace_bytes = ws.data_ptr(), need.value
            descs.append(d)
            keep += [xv, wp, wop, sc_d, b_d, bo_d, om, up, ws, out]
            res.append(up[3] if up is not None else out)
        arr = (_lib.DcnDesc * 2)(*descs)
        _lib.check(lib.ct_dcn_v2_group(arr, 2, _lib.CT_DCN_MAIN, _lib.stream_ptr()), 'main')
        _lib.check(lib.ct_dcn_v2_group(arr, 2, _lib.CT_DCN_FINISH, _lib.stream_ptr()), 'finish')
        torch.cuda.synchronize()
        outs[galgo] = [r.to_nchw().cpu() for r in res]
    for got, ref, y in zip(outs[xalgo], outs[43264], want):
        _close(got, y, msg='16-pixel group vs oracle')
        _close(got, ref, atol=2e-5, rtol=2e-5, msg='16- vs 32-pixel group')


@pytest.mark.parametrize('switch,value,code', [('DCN_TILE16', 600, '41664'), ('DCN_M32', True, '53264')])
def test_frame_plan_on_16_pixel_tiles_matches_the_oracle(device, monkeypatch, switch, value, code):
    """the headline configuration, one stream, with every MAIN launch below 600 workgroups on 16-pixel tiles: three frames
    through the whole path (forward, decode, association) against the CPU oracle at the full-size bar, and the plan
    really uses the shape"""
    from _parity import run_config
    from centertrack_amd import model as M
    monkeypatch.setattr(M, switch, value)
    checks, swaps, det = run_config('mot17_512', 1, 3, on_threshold_tie='stop')
    sigs = [M.DLASegHIP.plan_signature(p) for p in det.model._plans.values()]
    assert sigs and all(code in g for g in sigs)
    assert checks[0].frames >= 2
