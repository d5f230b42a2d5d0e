"""CPU: the oracle/ restatement reproduces the golden vectors that
tests/golden/make_golden.py generated from the reference's own modules."""
import json
import os

import numpy as np
import pytest
import torch

import scenarios as S
from centertrack_amd import weights as W
from oracle import decode as odecode
from oracle import detector as odet
from oracle import dla34
from oracle import image as oimage
from oracle import post_process as opost
from oracle import tracker as otracker


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize('name,heads,shape', [('mot', W.MOT_HEADS, (1, 64, 96)),
                                              ('nusc', W.NUSC_HEADS, (2, 64, 64))])
def test_model_forward_matches_reference(golden_dir, name, heads, shape):
    g = _load(golden_dir, 'model_forward.npz')
    sd = W.make_synthetic_state_dict(heads, seed=317)
    wsum = sum(float(v.double().abs().sum()) for v in sd.values())
    assert wsum == pytest.approx(float(g[name + '.wsum']), rel=1e-12), 'weight generator drifted'
    x, pre, hm = W.synthetic_inputs(*shape, seed=317)
    xsum = float(x.double().abs().sum() + pre.double().abs().sum() + hm.double().abs().sum())
    assert xsum == pytest.approx(float(g[name + '.xsum']), rel=1e-12)
    with torch.no_grad():
        y = dla34.forward(x, pre, hm, sd, heads)[-1]
        y1 = dla34.forward(x, pre, None, sd, heads)[-1]
    for k in heads:
        ref = g['%s.%s' % (name, k)]
        # same torch CPU ops in the same order; allow a few ulp for thread-count effects
        np.testing.assert_allclose(y[k].numpy(), ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y1['hm'].numpy(), g[name + '_nohm.hm'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('case', S.decode_cases(), ids=lambda c: c['name'])
def test_decode_matches_reference(golden_dir, case):
    g = _load(golden_dir, 'decode.npz')
    maps = S.make_head_maps(case)
    ret = odecode.generic_decode({k: v.clone() for k, v in maps.items()}, K=case['K'])
    keys = [k[len(case['name']) + 1:] for k in g.files if k.startswith(case['name'] + '.')]
    assert sorted(keys) == sorted(ret.keys())
    for k in keys:
        np.testing.assert_array_equal(ret[k].numpy(), g['%s.%s' % (case['name'], k)], err_msg=k)


def test_post_process_matches_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'post_process.json')))
    for case in S.postprocess_cases():
        r = opost.generic_post_process(case['out_thresh'], {k: v.copy() for k, v in case['dets'].items()},
                                       [case['c']], [case['s']], case['h'], case['w'], [case['calib']])[0]
        ref = g[case['name']]
        assert len(r) == len(ref)
        for a, b in zip(r, ref):
            assert sorted(a.keys()) == sorted(b.keys())
            for k in a:
                np.testing.assert_array_equal(np.asarray(a[k], np.float64), np.asarray(b[k], np.float64),
                                              err_msg='%s.%s' % (case['name'], k))


def _np_dets(dets):
    return [{k: (np.array(v, np.float32) if isinstance(v, list) else v) for k, v in d.items()}
            for d in dets]


def test_tracker_matches_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'tracker.json')))
    for seq in S.tracker_sequences():
        o = seq['opt']
        tr = otracker.Tracker(o['new_thresh'], o['max_age'], o['hungarian'], o['public_det'])
        tr.init_track([dict(d) for d in seq.get('pre_dets', [])])
        for t, fr in enumerate(seq['frames']):
            ret = tr.step(_np_dets(fr['dets']), fr.get('public_det'))
            got = [{'tracking_id': int(x['tracking_id']), 'age': int(x['age']), 'active': int(x['active']),
                    'score': float(x['score']), 'class': int(x['class'])} for x in ret]
            assert got == g[seq['name']][t], (seq['name'], t)


def test_pre_hm_matches_reference(golden_dir):
    g = _load(golden_dir, 'pre_hm.npz')
    for case in S.pre_hm_cases():
        opt = odet.default_opt(pre_thresh=case['pre_thresh'], flip_test=case['flip_test'], tracking=False)
        hm, inds = odet.render_pre_hm(opt, case['tracks'], case['meta'])
        np.testing.assert_array_equal(hm.numpy(), g[case['name'] + '.hm'])
        np.testing.assert_array_equal(inds.numpy(), g[case['name'] + '.inds'])
        m = case['meta']
        np.testing.assert_array_equal(
            oimage.get_affine_transform(m['c'], m['s'], 0, [m['inp_width'], m['inp_height']]),
            g[case['name'] + '.trans_input'])
        np.testing.assert_array_equal(
            oimage.get_affine_transform(m['c'], m['s'], 0, [m['out_width'], m['out_height']], inv=1),
            g[case['name'] + '.trans_output_inv'])


def test_e2e_detector_matches_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'e2e_mot.json')))
    cfg = S.e2e_config()
    sd = S.e2e_state_dict(cfg)
    opt = odet.default_opt(track_thresh=cfg['track_thresh'], pre_thresh=cfg['pre_thresh'],
                           input_h=cfg['H'], input_w=cfg['W'])
    det = odet.Detector(opt, sd, cfg['heads'])
    for t, (images, meta) in enumerate(S.e2e_frames(cfg)):
        res = det.run(images, meta)
        ref = g['frames'][t]
        assert [int(r['tracking_id']) for r in res] == [int(r['tracking_id']) for r in ref], t
        assert [int(r['class']) for r in res] == [int(r['class']) for r in ref]
        for a, b in zip(res, ref):
            for k in ('score', 'ct', 'bbox', 'tracking'):
                np.testing.assert_allclose(np.asarray(a[k], np.float64), np.asarray(b[k], np.float64),
                                           rtol=1e-4, atol=1e-4, err_msg='frame %d %s' % (t, k))
            assert int(a['age']) == int(b['age']) and int(a['active']) == int(b['active'])


@pytest.mark.parametrize('case', S.e2e_mode_cases(), ids=lambda c: c['name'])
def test_e2e_modes_match_reference(golden_dir, case):
    """reference Detector.run (tests/golden/e2e_modes.json) in every mode a BASELINE configuration or the MOT protocol
    uses -- T = 16, Hungarian, max_age 2, public detections with pre_dets / cur_dets, --flip_test, tracking,ddd with calib,
    80 classes -- against the oracle detector: ids / classes / ages identical, values to 1e-4"""
    g = json.load(open(os.path.join(golden_dir, 'e2e_modes.json')))[case['name']]
    sd = S.e2e_mode_state_dict(case, S.e2e_mode_calibration(case, golden_dir))
    opt = odet.default_opt(input_h=case['H'], input_w=case['W'], num_classes=case['heads']['hm'], **case['opt'])
    det = odet.Detector(opt, sd, case['heads'])
    assert len(g) == case['T']
    for t, (images, meta) in enumerate(S.e2e_mode_frames(case)):
        res = det.run(images, meta)
        ref = g[t]
        assert [int(r['tracking_id']) for r in res] == [int(r['tracking_id']) for r in ref], t
        assert [int(r['class']) for r in res] == [int(r['class']) for r in ref]
        for a, b in zip(res, ref):
            assert int(a['age']) == int(b['age']) and int(a['active']) == int(b['active'])
            for k in ('score', 'ct', 'bbox', 'tracking', 'dep', 'dim', 'alpha', 'loc', 'rot_y'):
                assert (k in a) == (k in b), k
                if k in b:
                    np.testing.assert_allclose(np.asarray(a[k], np.float64).reshape(-1), np.asarray(b[k], np.float64).reshape(-1),
                                               rtol=1e-4, atol=2e-4, err_msg='%s frame %d %s' % (case['name'], t, k))


@pytest.mark.parametrize('name,h,w', [('mot_512', 512, 512), ('mot_544x960', 544, 960)])
def test_model_forward_full_size_matches_reference(golden_dir, name, h, w):
    """the oracle forward pinned to the reference's DLASeg at the sizes that are BENCHMARKED (model_forward.npz holds
    64 x 96 / 64 x 64 only): 512 x 512 and the reference's own 544 x 960 with its ragged 17 x 30 deep maps"""
    g = _load(golden_dir, 'model_forward_full.npz')
    heads = W.MOT_HEADS
    sd = W.make_synthetic_state_dict(heads, seed=317)
    x, pre, hm = W.synthetic_inputs(1, h, w, seed=317)
    with torch.no_grad():
        y = dla34.forward(x, pre, hm, sd, heads)[-1]
    for k in heads:
        got = y[k].numpy() if k == 'hm' else y[k][:, :, ::2, ::2].numpy()
        np.testing.assert_allclose(got, g['%s.%s' % (name, k)], rtol=1e-5, atol=2e-5, err_msg=k)


def test_pose_flip_matches_reference(golden_dir):
    """oracle mirror_joints == the reference's flip_lr (hm_hp) / flip_lr_off (hps), model/utils.py:33-50"""
    from oracle import detector as odet
    g = _load(golden_dir, 'pose_flip.npz')
    x = S.pose_flip_inputs()
    np.testing.assert_array_equal(odet.mirror_joints(x['hm_hp'], odet.COCO_FLIP_IDX, False).numpy(), g['hm_hp'])
    np.testing.assert_array_equal(odet.mirror_joints(x['hps'], odet.COCO_FLIP_IDX, True).numpy(), g['hps'])
