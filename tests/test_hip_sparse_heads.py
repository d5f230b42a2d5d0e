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
