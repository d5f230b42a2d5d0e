"""GPU: sparse heads (opt.sparse_heads, ct_sparse_heads_desc; round 5, opt-in).  The regression heads are evaluated at the
K winners of the decode only -- generic_decode reads nothing else of them (decode.py:99-180) -- so every packed row must
be what the dense path gives: the same winners (the hm map is the same launch), head values within fp32 rounding of the
dense Winograd launch (direct convolution here), and, against the CPU oracle at full size, the same assertions as the
dense full-size tests (tests/_parity.py: ranks / classes / ids identical, values within 1e-3 on the output grid)."""
import numpy as np
import pytest
import torch

from _parity import run_config

pytestmark = pytest.mark.gpu


def _pair(name, streams, **kw):
    import scenarios as S
    from _parity import calibrated_state_dict
    from centertrack_amd.detector import StreamDetector, default_opt
    from centertrack_amd.image import make_meta
    from centertrack_amd.model import DLASegHIP
    cfg = S.CONFIGS[name]
    heads = S.HEAD_SETS[cfg['heads']]
    sd = calibrated_state_dict(name, heads)
    dets = []
    kw = dict(kw)
    flip = kw.pop('flip_test', cfg['flip'])
    for sparse in (False, True):
        opt = default_opt(heads, track_thresh=cfg['track_thresh'], pre_thresh=cfg['pre_thresh'], sparse_heads=sparse,
                          flip_test=flip, **kw)
        model = DLASegHIP(heads)
        model.load_state_dict(sd)
        dets.append(StreamDetector(opt, model=model, num_streams=streams))
    return cfg, dets[0], dets[1], make_meta(cfg['H'], cfg['W'], 2 * cfg['H'], 2 * cfg['W'])


@pytest.mark.parametrize('name,streams,kw', [('mot17_512', 1, {}), ('mot17_512', 3, {'zero_tracking': True}),
                                              ('nusc_800x448', 2, {}), ('coco_512', 2, {}), ('mot17_544x960', 1, {}),
                                              ('kitti_1280x384', 2, {}), ('nusc_800x448', 1, {'flip_test': True})],
                         ids=['mot', 'mot_x3_zero_tracking', 'nusc_3d_heads', 'coco_80_classes', 'mot_544x960',
                              'kitti_flip_test', 'nusc_3d_heads_flip_test'])
def test_sparse_rows_equal_dense_rows(device, name, streams, kw):
    from _parity import scrolled_stream
    cfg, dense, sparse, meta = _pair(name, streams, **kw)
    assert sparse.sparse and not dense.sparse
    frames = [scrolled_stream(cfg['H'], cfg['W'], 4, 400 + 10 * s) for s in range(streams)]
    for t in range(4):
        x = torch.cat([frames[s][t] for s in range(streams)], 0)
        rd = dense.step(x, [dict(meta) for _ in range(streams)])
        rs = sparse.step(x, [dict(meta) for _ in range(streams)])
        a, b = dense.last_dets, sparse.last_dets
        assert sorted(a) == sorted(b)
        for k in ('scores', 'clses', 'xs', 'ys'):                    # the winners: same hm launch, same selection
            np.testing.assert_array_equal(a[k], b[k], err_msg='frame %d %s' % (t, k))
        for k in a:
            if k not in ('scores', 'clses', 'xs', 'ys', 'cts'):
                scale = max(1.0, float(np.abs(a[k]).max()))
                np.testing.assert_allclose(b[k], a[k], rtol=0, atol=2e-4 * scale, err_msg='frame %d %s' % (t, k))
        for s in range(streams):
            assert [int(r['tracking_id']) for r in rs[s]] == [int(r['tracking_id']) for r in rd[s]], (t, s)
    if kw.get('zero_tracking'):
        assert float(np.abs(sparse.last_dets['tracking']).max()) == 0.0
    plan = sparse._ctx['plan']
    names = [l.name for l in plan['launches'] if l.fn == 'heads' or l.name.startswith('heads.')]
    assert plan['sparse'] is not None and all('wh' not in n and 'tracking' not in n for n in names), names


# (streams picked like the dense full-size tests': no oracle score within 1e-5 of a threshold -- checked on the CPU, min 3.6e-4)
@pytest.mark.parametrize('name,streams,T,seed0', [('mot17_512', 1, 8, 331), ('nusc_800x448', 2, 3, 324), ('coco_512', 2, 3, 324),
                                                   ('kitti_1280x384', 2, 2, 317 + 7)])
def test_sparse_heads_match_oracle_at_full_size(device, name, streams, T, seed0):
    checks, swaps, det = run_config(name, streams, T, seed0=seed0, sparse_heads=True)
    assert det.sparse
    assert sum(c.frames for c in checks) == T * streams


@pytest.mark.parametrize('name,streams,sample,T', [('mot17_512', 16, (0, 8, 15), 3), ('nusc_800x448', 8, (0, 7), 3),
                                                    ('kitti_1280x384', 4, (0, 3), 2)])
def test_sparse_heads_on_the_benchmarked_many_stream_plans(device, name, streams, sample, T):
    """the many-stream launch plans (other conv tiles, Winograd raw-sum offset convs; tests/test_hip_plans.py) with sparse
    heads against the oracle: streams nobody picked, so a stream that meets a threshold tie is compared up to that frame"""
    from _parity import RANK_TIE_UNPICKED
    checks, swaps, det = run_config(name, streams, T, sample=sample, on_threshold_tie='stop', min_tracks=5,
                                    rank_tie=RANK_TIE_UNPICKED, sparse_heads=True)
    assert det.sparse
    assert sum(c.frames for c in checks) >= (2 * T * len(sample) + 2) // 3


def test_sparse_heads_are_refused_where_they_do_not_apply(device):
    import scenarios as S
    from centertrack_amd import weights as W
    from centertrack_amd.detector import StreamDetector, default_opt
    from centertrack_amd.model import DLASegHIP
    heads = W.KITTI_HEADS
    model = DLASegHIP(heads)
    model.load_state_dict(W.make_synthetic_state_dict(heads, seed=3))
    det = StreamDetector(default_opt(heads, flip_test=True, sparse_heads=True), model=model, num_streams=1)
    assert det.sparse                            # flip_test: hm merged as a map, averaged heads evaluated in both images
    pose = StreamDetector(default_opt(W.POSE_HEADS, sparse_heads=True), model=None if False else DLASegHIP(W.POSE_HEADS),
                          num_streams=1)
    assert not pose.sparse


@pytest.mark.parametrize('B,h,w,K,names,flip', [(1, 24, 40, 37, ('reg', 'wh', 'tracking'), False),
                                               (3, 17, 30, 100, ('tracking', 'ltrb_amodal', 'dep', 'rot', 'dim', 'amodel_offset'), False),
                                               (2, 16, 24, 20, ('reg', 'wh', 'dep', 'dim', 'amodel_offset', 'tracking'), True),
                                               (1, 32, 32, 100, ('wh', 'ltrb', 'nuscenes_att', 'velocity'), False),
                                               (2, 40, 40, 300, ('reg', 'wh', 'tracking'), False)],
                         ids=['odd_K', 'ragged_map_3d_heads_x3', 'flip_test', 'ltrb_att_velocity', 'K_300'])
def test_ct_decode_sparse_against_torch_convolutions(device, B, h, w, K, names, flip):
    """ct_decode with ct_sparse_heads_desc at the C-ABI level, on random feature maps and random head weights, against
    plain torch fp32 convolutions evaluated densely on the CPU and gathered at the winners: ragged maps, K not a
    multiple of 16, winners on the border (zero padding), 8-channel heads, the dep transform, flip_test averaging with the
    x channels of amodel_offset negated."""
    import torch.nn.functional as F
    from centertrack_amd import _lib, ops
    g = torch.Generator().manual_seed(11 + K)
    NB = 2 * B if flip else B
    feat = torch.randn((NB, 64, h, w), generator=g)
    hm = torch.rand((B, 2, h, w), generator=g)
    hm[:, :, 0, :] += 0.5                                    # winners on the top border too
    hm = hm.clamp(0, 0.999)
    heads = []
    dense = {}
    for n in names:
        c = _lib.HEAD_CH[n]
        w1 = torch.randn((256, 64, 3, 3), generator=g) * 0.05
        b1 = torch.randn((256,), generator=g) * 0.1
        w2 = torch.randn((c, 256), generator=g) * 0.05
        b2 = torch.randn((c,), generator=g) * 0.1
        heads.append((n, w1, b1, w2, b2))
        v = F.conv2d(F.relu(F.conv2d(feat, w1, b1, padding=1)), w2.view(c, 256, 1, 1), b2)
        if n == 'dep':
            v = (1.0 / (torch.sigmoid(v) + 1e-6) - 1.0) * 2.0
        if flip:
            a, bb = v[:B], torch.flip(v[B:], [3])
            if n in ('wh', 'dep', 'dim'):
                v = (a + bb) / 2
            elif n == 'amodel_offset':
                bb = bb.clone()
                bb[:, 0::2] *= -1
                v = (a + bb) / 2
            else:
                v = a
        dense[n] = v.contiguous()
    fv = ops.view_from_nchw(feat.to(device))
    sparse = {'feat': fv, 'flip': flip, 'depth_scale': 2.0,
              'heads': [(n, ops.pack_weight(w1.to(device)), b1.to(device), w2.to(device).contiguous(), b2.to(device))
                        for n, w1, b1, w2, b2 in heads]}
    dec = ops.Decoder(hm.to(device), {}, K, sparse=sparse)
    got = dec.unpack(dec.run().cpu().numpy())
    ref = ops.Decoder(hm.to(device), {n: v.to(device) for n, v in dense.items()}, K)
    want = ref.unpack(ref.run().cpu().numpy())
    assert sorted(got) == sorted(want)
    for k in want:
        if k in ('scores', 'clses', 'xs', 'ys', 'cts'):
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        else:
            np.testing.assert_allclose(got[k], want[k], rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(want[k]).max())), err_msg=k)
    assert (want['ys'] == 0).any()                            # border winners were part of it
