"""Seeded synthetic workloads shared by the golden generator, the parity tests and
bench.py: head maps for decode, detection lists for post-process / tracker, and the
end-to-end stream configs of BASELINE.json.  Pure numpy/torch CPU
generators; MT19937 / torch CPU generators are stable across machines, so the same seed
reproduces the same inputs on the GPU box."""
from collections import OrderedDict

import numpy as np
import torch

from centertrack_amd import weights as W

HEAD_SETS = {'mot': W.MOT_HEADS, 'kitti': W.KITTI_HEADS, 'coco': W.COCO_HEADS, 'nusc': W.NUSC_HEADS}

# BASELINE.json configs -> (head set, input H, input W, streams/batch, flip_test, track_thresh)
CONFIGS = OrderedDict([
    ('mot17_512', dict(heads='mot', H=512, W=512, B=1, flip=False, track_thresh=0.4, pre_thresh=0.5)),
    ('kitti_1280x384', dict(heads='kitti', H=384, W=1280, B=4, flip=True, track_thresh=0.4, pre_thresh=0.4)),
    ('coco_512', dict(heads='coco', H=512, W=512, B=32, flip=False, track_thresh=0.3, pre_thresh=0.3)),
    ('nusc_800x448', dict(heads='nusc', H=448, W=800, B=16, flip=False, track_thresh=0.1, pre_thresh=0.1)),
    # not a BASELINE.json line: the reference's own MOT input size (src/lib/dataset/datasets/mot.py:15, the size
    # `test.py tracking --dataset mot` runs at and readme/MODEL_ZOO.md:16-20 quotes its time on); deep maps 17 x 30
    ('mot17_544x960', dict(heads='mot', H=544, W=960, B=1, flip=False, track_thresh=0.4, pre_thresh=0.5)),
])


# ----------------------------------------------------------------------------- decode
def decode_cases():
    return [
        dict(name='mot_32x40', heads=W.MOT_HEADS, B=1, h=32, w=40, K=100, seed=1),
        dict(name='kitti_24x80', heads=W.KITTI_HEADS, B=2, h=24, w=80, K=100, seed=2),
        dict(name='coco_16x16', heads=W.COCO_HEADS, B=1, h=16, w=16, K=100, seed=3),
        dict(name='nusc_28x50', heads=W.NUSC_HEADS, B=3, h=28, w=50, K=64, seed=4),
        dict(name='mot_128x128', heads=W.MOT_HEADS, B=1, h=128, w=128, K=100, seed=5),
        dict(name='pose_32x48', heads=W.POSE_HEADS, B=1, h=32, w=48, K=100, seed=6),   # (the reference's pose branch only runs at batch 1)
        # pose without any box head (decode.py:60-71: joints gated by the extent of the regressed key points) and with
        # an ltrb_amodal head next to wh (the gate stays the wh box, ret['bboxes'] is the amodal one, decode.py:159)
        dict(name='pose_nobox_24x32', heads=OrderedDict([(k, v) for k, v in W.POSE_HEADS.items() if k != 'wh']),
             B=1, h=24, w=32, K=60, seed=7, hps_std=1.0),
        dict(name='pose_amodal_24x32', heads=OrderedDict(list(W.POSE_HEADS.items()) + [('ltrb_amodal', 4)]),
             B=1, h=24, w=32, K=60, seed=8),
        # round 5: the output grids of the BENCHMARKED configurations (BASELINE configs 3-5 and the reference's own MOT size):
        # 80 classes x 16 384 pixels (the multi-slice selection), 112 x 200 with the 3D heads, the wide 96 x 320 KITTI grid,
        # the ragged 136 x 240 grid of 544 x 960 (seeds picked so that no two of an image's K winners score exactly the same:
        # torch.topk's order among exact ties is unspecified, the generator refuses such data)
        dict(name='coco_128x128', distinct_hm=True, heads=W.COCO_HEADS, B=2, h=128, w=128, K=100, seed=9),
        dict(name='nusc_112x200', distinct_hm=True, heads=W.NUSC_HEADS, B=2, h=112, w=200, K=100, seed=10),
        dict(name='kitti_96x320', distinct_hm=True, heads=W.KITTI_HEADS, B=2, h=96, w=320, K=100, seed=11),
        dict(name='mot_136x240', distinct_hm=True, heads=W.MOT_HEADS, B=1, h=136, w=240, K=100, seed=12),
    ]


def make_head_maps(case):
    """Post-``_sigmoid_output`` head maps: hm in (0,1) with distinct values (smooth bumps +
    noise so that the 3x3 NMS keeps ~10 % of the pixels), regression heads ~ N(0,s)."""
    g = torch.Generator().manual_seed(1000 + case['seed'])
    B, h, w = case['B'], case['h'], case['w']
    out = OrderedDict()
    for name, c in case['heads'].items():
        if name in ('hm', 'hm_hp'):
            if case.get('distinct_hm'):
                # millions of U(0,1) draws rounded to fp32 collide among the top K; ranks of a random permutation do not
                n = B * c * h * w
                v = ((torch.randperm(n, generator=g).double() + 0.5) / n).view(B, c, h, w)
            else:
                v = torch.rand((B, c, h, w), generator=g, dtype=torch.float64)
            out[name] = (v ** 2 * 0.98 + 0.001).float()
        elif name in ('reg', 'hp_offset'):
            out[name] = torch.rand((B, c, h, w), generator=g, dtype=torch.float64).float()
        elif name == 'wh' and 'hps' in case['heads']:     # pose: boxes large enough that joints snap to heat-map peaks
            out[name] = (torch.randn((B, c, h, w), generator=g, dtype=torch.float64) * 4 + 11).float()
        elif name == 'wh':
            out[name] = (torch.randn((B, c, h, w), generator=g, dtype=torch.float64) * 4 + 3).float()
        elif name == 'hps':
            out[name] = (torch.randn((B, c, h, w), generator=g, dtype=torch.float64) * case.get('hps_std', 3.0)).float()
        elif name == 'dep':
            out[name] = (torch.rand((B, c, h, w), generator=g, dtype=torch.float64) * 60 + 1).float()
        else:
            out[name] = (torch.randn((B, c, h, w), generator=g, dtype=torch.float64) * 2).float()
    return out


# ----------------------------------------------------------------------- post-process
def _sorted_dets(rs, K, F_extra, n_above, thresh):
    scores = np.sort(rs.uniform(0.0, 1.0, size=K).astype(np.float32))[::-1].copy()
    scores[:n_above] = np.maximum(scores[:n_above], thresh + 0.01)
    scores[n_above:] = np.minimum(scores[n_above:], thresh - 0.01)
    scores = np.sort(scores)[::-1].copy()
    d = {'scores': scores[None]}
    for k, f in F_extra.items():
        d[k] = f(rs, K)[None]
    return d


def postprocess_cases():
    cases = []
    rs = np.random.RandomState(11)
    K = 40
    ext2d = {
        'clses': lambda r, k: r.randint(0, 3, size=k).astype(np.float32),
        'cts': lambda r, k: r.uniform(0, 100, size=(k, 2)).astype(np.float32),
        'tracking': lambda r, k: r.normal(0, 3, size=(k, 2)).astype(np.float32),
        'bboxes': lambda r, k: np.sort(r.uniform(0, 100, size=(k, 2, 2)), axis=1).reshape(k, 4).astype(np.float32),
    }
    cases.append(dict(name='mot2d', out_thresh=0.4, num_classes=3, h=96, w=128,
                      c=np.array([640., 360.], np.float32), s=1280.0, height=720, width=1280,
                      calib=np.array([[1200., 0, 640, 0], [0, 1200., 360, 0], [0, 0, 1, 0]], np.float32),
                      dets=_sorted_dets(rs, K, ext2d, 17, 0.4)))
    ext3d = dict(ext2d)
    ext3d.update({
        'dep': lambda r, k: r.uniform(2, 60, size=(k, 1)).astype(np.float32),
        'rot': lambda r, k: r.normal(0, 1, size=(k, 8)).astype(np.float32),
        'dim': lambda r, k: r.uniform(0.5, 4, size=(k, 3)).astype(np.float32),
        'amodel_offset': lambda r, k: r.normal(0, 1, size=(k, 2)).astype(np.float32),
    })
    cases.append(dict(name='nusc3d', out_thresh=0.1, num_classes=10, h=112, w=200,
                      c=np.array([800., 450.], np.float32), s=1600.0, height=900, width=1600,
                      calib=np.array([[1266.4, 0, 816.3, 0], [0, 1266.4, 491.5, 0], [0, 0, 1, 0]], np.float32),
                      dets=_sorted_dets(rs, K, ext3d, 25, 0.1)))
    # s given as an array (keep_res path, detector.py:195-199)
    cases.append(dict(name='keepres', out_thresh=0.3, num_classes=1, h=96, w=320,
                      c=np.array([620., 187.], np.float32), s=np.array([1280., 384.], np.float32),
                      height=375, width=1242,
                      calib=np.array([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.2], [0, 0, 1, 0.003]], np.float32),
                      dets=_sorted_dets(rs, K, ext2d, 9, 0.3)))
    # key points (tracking,multi_pose): 17 joints per detection, post_process.py:51-54
    extpose = dict(ext2d)
    extpose['hps'] = lambda r, k: r.uniform(-5, 130, size=(k, 34)).astype(np.float32)
    cases.append(dict(name='pose2d', out_thresh=0.3, num_classes=1, h=128, w=128,
                      c=np.array([320., 240.], np.float32), s=640.0, height=480, width=640,
                      calib=np.array([[1200., 0, 320, 0], [0, 1200., 240, 0], [0, 0, 1, 0]], np.float32),
                      dets=_sorted_dets(rs, K, extpose, 12, 0.3)))
    return cases


def pose_flip_inputs():
    """flipped-image halves of the pose heads for the flip_lr / flip_lr_off parity check: hm_hp [1,17,6,10] and
    hps [1,34,6,10]"""
    g = torch.Generator().manual_seed(77)
    return {'hm_hp': torch.rand((1, 17, 6, 10), generator=g), 'hps': torch.randn((1, 34, 6, 10), generator=g)}


# ---------------------------------------------------------------------------- tracker
def _det(score, cls, ct, tracking, size):
    ct = [float(ct[0]), float(ct[1])]
    half = size / 2.0
    return {'score': float(score), 'class': int(cls), 'ct': ct,
            'tracking': [float(tracking[0]), float(tracking[1])],
            'bbox': [ct[0] - half, ct[1] - half, ct[0] + half, ct[1] + half]}


def tracker_sequences():
    seqs = []
    base = dict(new_thresh=0.4, max_age=-1, hungarian=False, public_det=False)
    # A: three objects moving +8 px/frame; tracking points back by -8; births and deaths
    frames = []
    for t in range(5):
        dets = []
        for i, (x0, y0, sc) in enumerate([(100, 100, 0.9), (300, 120, 0.8), (200, 300, 0.7)]):
            if i == 1 and t >= 3:
                continue                                     # object 1 leaves at t=3
            dets.append(_det(sc - 0.01 * t, 1, (x0 + 8 * t, y0 + 2 * t), (-8, -2), 40))
        if t >= 2:
            dets.append(_det(0.95, 1, (500 + 8 * t, 400), (-8, 0), 50))   # born at t=2, top score
        dets.sort(key=lambda d: -d['score'])
        frames.append({'dets': dets})
    seqs.append(dict(name='basic', opt=base, frames=frames))
    # B: gating -- far jump, class change, low score, size gate by track vs item
    frames = [
        {'dets': [_det(0.9, 1, (100, 100), (0, 0), 30), _det(0.8, 2, (200, 100), (0, 0), 30),
                  _det(0.7, 1, (300, 100), (0, 0), 8), _det(0.35, 1, (400, 100), (0, 0), 30)]},
        {'dets': [_det(0.9, 1, (160, 100), (0, 0), 30),     # jumped 60 px  -> new id
                  _det(0.8, 1, (200, 100), (0, 0), 30),     # class changed -> new id
                  _det(0.7, 1, (307, 100), (0, 0), 30),     # 49 > track size 64? no: 49<64 & <900
                  _det(0.6, 1, (400, 100), (0, 0), 30)]},   # previous was below new_thresh
        {'dets': [_det(0.9, 1, (163, 104), (0, 0), 4),      # dist 25 > item size 16 -> new id
                  _det(0.5, 1, (200, 100), (0, 0), 30)]},
    ]
    seqs.append(dict(name='gating', opt=base, frames=frames))
    # C: max_age keeps an unmatched track alive
    opt_c = dict(base, max_age=2)
    frames = [{'dets': [_det(0.9, 1, (100, 100), (0, 0), 40), _det(0.8, 1, (300, 100), (0, 0), 40)]},
              {'dets': [_det(0.9, 1, (102, 100), (-2, 0), 40)]},
              {'dets': [_det(0.9, 1, (104, 100), (-2, 0), 40), _det(0.8, 1, (301, 100), (-1, 0), 40)]},
              {'dets': []},
              {'dets': [_det(0.8, 1, (302, 100), (-1, 0), 40)]},
              {'dets': [_det(0.8, 1, (303, 100), (-1, 0), 40)]}]
    seqs.append(dict(name='max_age', opt=opt_c, frames=frames))
    # D: greedy vs Hungarian differ when the best det of a track is taken first
    frames = [{'dets': [_det(0.9, 1, (100, 100), (0, 0), 60), _det(0.8, 1, (120, 100), (0, 0), 60)]},
              {'dets': [_det(0.9, 1, (112, 100), (0, 0), 60), _det(0.8, 1, (131, 100), (0, 0), 60)]},
              {'dets': [_det(0.9, 1, (125, 100), (0, 0), 60), _det(0.8, 1, (140, 100), (0, 0), 60)]}]
    seqs.append(dict(name='greedy_cross', opt=base, frames=frames))
    seqs.append(dict(name='hungarian_cross', opt=dict(base, hungarian=True), frames=frames))
    # E: public detection mode (tracker.py:83-101)
    frames = [{'dets': [_det(0.9, 1, (100, 100), (0, 0), 40), _det(0.8, 1, (300, 100), (0, 0), 40)],
               'public_det': [{'ct': [101., 101.]}, {'ct': [500., 500.]}]},
              {'dets': [_det(0.9, 1, (102, 100), (-2, 0), 40), _det(0.85, 1, (300, 102), (0, -2), 40),
                        _det(0.7, 1, (400, 300), (0, 0), 40)],
               'public_det': [{'ct': [299., 101.]}, {'ct': [402., 301.]}]}]
    seqs.append(dict(name='public_det', opt=dict(base, public_det=True), frames=frames))
    # F: random stress, many dets, random motion, three classes, inherited init tracks
    rs = np.random.RandomState(7)
    objs = [dict(p=rs.uniform(50, 900, 2), v=rs.normal(0, 6, 2), c=int(rs.randint(1, 4)),
                 s=float(rs.uniform(15, 80))) for _ in range(40)]
    frames = []
    for t in range(8):
        dets = []
        for o in objs:
            if rs.uniform() < 0.1:
                continue
            o['p'] = o['p'] + o['v']
            noise = rs.normal(0, 2.0, 2)
            dets.append(_det(float(np.float32(rs.uniform(0.2, 1.0))), o['c'], np.float32(o['p']),
                             np.float32(-o['v'] + noise), o['s']))
        dets.sort(key=lambda d: -d['score'])
        frames.append({'dets': dets})
    pre = [dict(_det(0.9, 1, (60, 60), (0, 0), 30)), dict(_det(0.3, 2, (90, 90), (0, 0), 30))]
    for p in pre:
        del p['ct']                                           # init_track derives ct from bbox
    seqs.append(dict(name='random', opt=base, frames=frames, pre_dets=pre))
    seqs.append(dict(name='random_hungarian', opt=dict(base, hungarian=True, max_age=3), frames=frames))
    return seqs


# ----------------------------------------------------------------------------- pre-hm
def pre_hm_cases():
    from centertrack_amd.image import make_meta
    cases = []
    rs = np.random.RandomState(5)
    for name, (ih, iw, oh, ow), flip in [('mot', (720, 1280, 512, 512), False),
                                        ('kitti', (375, 1242, 384, 1280), True),
                                        ('small', (96, 128, 96, 128), False)]:
        meta = make_meta(oh, ow, ih, iw, down_ratio=4)
        tracks = []
        for i in range(12):
            cx, cy = rs.uniform(-20, iw + 20), rs.uniform(-20, ih + 20)
            bw, bh = rs.uniform(2, iw / 3), rs.uniform(2, ih / 3)
            tracks.append({'score': float(rs.uniform(0.2, 1.0)), 'active': int(rs.randint(0, 3)),
                           'bbox': np.array([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], np.float32)})
        cases.append(dict(name=name, meta=meta, tracks=tracks, pre_thresh=0.5, flip_test=flip))
    return cases


# -------------------------------------------------------------------------------- e2e
def e2e_config():
    return dict(heads=W.MOT_HEADS, H=128, W=160, T=4, seed=317, hm_gain=14.0,
                orig_h=360, orig_w=480,
                ref_args=['--pre_hm', '--ltrb_amodal', '--track_thresh', '0.4', '--pre_thresh', '0.5'],
                track_thresh=0.4, pre_thresh=0.5)


def e2e_state_dict(cfg):
    """Synthetic weights of the e2e stream: hm output layer scaled so scores spread over
    (0,1), and the amodal-box head biased to ~6x6-cell boxes so the tracker's size gate
    (dist^2 < box area, tracker.py:47-48) lets consecutive frames associate."""
    sd = W.make_synthetic_state_dict(cfg['heads'], seed=cfg['seed'], hm_gain=cfg['hm_gain'])
    sd['ltrb_amodal.2.bias'] = torch.tensor([-3.0, -3.0, 3.0, 3.0])
    return sd


def e2e_frames(cfg):
    """T frames of one synthetic stream: a fixed N(0,1) image scrolled by 4 input px/frame
    (so detections drift by one output cell) + the per-frame ``meta`` of pre_process."""
    from centertrack_amd.image import make_meta
    g = torch.Generator().manual_seed(cfg['seed'] + 7)
    base = torch.randn((3, cfg['H'], cfg['W'] + 4 * cfg['T']), generator=g, dtype=torch.float64).float()
    meta = make_meta(cfg['H'], cfg['W'], cfg['orig_h'], cfg['orig_w'], down_ratio=4)
    for t in range(cfg['T']):
        img = base[:, :, 4 * t:4 * t + cfg['W']].contiguous().unsqueeze(0)
        yield img, dict(meta)


# ------------------------------------------------------------------ e2e: the other modes of Detector.run (round 5)
def e2e_mode_cases():
    """Reference ``Detector.run`` goldens beyond the 4-frame MOT stream (VERDICT r4 "missing" 2): longer streams and every
    mode of detector.py:55-172 / tracker.py that a BASELINE configuration or the MOT protocol uses.  ``ref_args`` are the
    reference's own command-line flags (opts.py); ``opt`` the same settings for ``default_opt`` here / in the oracle."""
    base = dict(H=128, W=160, orig_h=360, orig_w=480, seed=317, hm_gain=14.0, box_cells=6.0, calibrated=False)
    mot = ['--pre_hm', '--ltrb_amodal', '--track_thresh', '0.4', '--pre_thresh', '0.5']
    cases = [
        dict(name='mot_t16', heads=W.MOT_HEADS, T=16, task='tracking', ref_args=mot,
             opt=dict(track_thresh=0.4, pre_thresh=0.5)),
        dict(name='mot_hungarian', heads=W.MOT_HEADS, T=8, task='tracking', ref_args=mot + ['--hungarian'],
             opt=dict(track_thresh=0.4, pre_thresh=0.5, hungarian=True)),
        # (own gain: at 14 two detections of frame 1 score EXACTLY the same saturated fp32 value, and the order torch.topk gives
        # exact ties is unspecified -- SURVEY.md App. D.1; the generator refuses such data)
        dict(name='mot_max_age2', heads=W.MOT_HEADS, T=10, task='tracking', ref_args=mot + ['--max_age', '2'], seed=317, hm_gain=10.0,
             opt=dict(track_thresh=0.4, pre_thresh=0.5, max_age=2)),
        # MOT public-detection protocol (test.py:88-107, tracker.py:83-101): births only next to a provided detection
        dict(name='mot_public', heads=W.MOT_HEADS, T=8, task='tracking', ref_args=mot + ['--public_det', '--load_results', 'x'],
             opt=dict(track_thresh=0.4, pre_thresh=0.5, public_det=True), public_grid=60),
        # KITTI heads + --flip_test (experiments/kitti_half.sh:5): batch of 2, merged by _flip_output (detector.py:311-332)
        dict(name='kitti_flip', heads=W.KITTI_HEADS, T=6, task='tracking', calibrated=True,
             ref_args=['--pre_hm', '--flip_test', '--track_thresh', '0.4'], opt=dict(track_thresh=0.4, flip_test=True)),
        # nuScenes tracking,ddd (experiments/nuScenes_3Dtracking.sh): 3D fields through generic_post_process with calib
        dict(name='nusc_ddd', heads=W.NUSC_HEADS, T=6, task='tracking,ddd', calibrated=True, W=224,
             ref_args=['--pre_hm', '--track_thresh', '0.1'], opt=dict(track_thresh=0.1)),
        # 80 classes (COCO): the class-major top-K over 80 maps, then over 8000 candidates
        dict(name='coco80', heads=W.COCO_HEADS, T=4, task='tracking', calibrated=True,
             ref_args=['--pre_hm', '--track_thresh', '0.3'], opt=dict(track_thresh=0.3)),
        # the BENCHMARKED size: BASELINE configs[1] = MOT heads at 512 x 512, the calibrated weights of the full-size parity
        # tests (tests/golden/hm_calibration.json: ~40 detections per frame, scores spread), T = 8
        dict(name='mot_512_full', heads=W.MOT_HEADS, T=8, task='tracking', ref_args=mot, H=512, W=512, orig_h=1024, orig_w=1024,
             opt=dict(track_thresh=0.4, pre_thresh=0.5), calibration_of='mot17_512', frames_seed=359),
    ]
    return [dict(base, **c) for c in cases]


def e2e_mode_calibration(case, golden_dir):
    """the hm calibration entry of an e2e mode case (None for the single-class MOT cases at small size)"""
    import json
    import os
    if case.get('calibration_of'):
        with open(os.path.join(golden_dir, 'hm_calibration.json')) as f:
            return json.load(f)[case['calibration_of']]
    with open(os.path.join(golden_dir, 'e2e_modes_calibration.json')) as f:
        return json.load(f).get(case['name'])


def e2e_mode_state_dict(case, calibration=None):
    """seeded weights of an e2e mode case: hm output layer scaled (scores spread over (0,1)), boxes of ``box_cells``
    output cells so that the tracker's size gate lets consecutive frames associate.  Multi-class heads take the
    per-class scale / bias of tests/golden/e2e_modes_calibration.json (``calibration`` = that file's entry for the case;
    made by tests/golden/make_golden.py from frame 0: random weights would let ONE class supply every detection)"""
    if calibration is not None:
        sd = W.make_synthetic_state_dict(case['heads'], seed=case['seed'], hm_gain=1.0)
        sc = torch.tensor(calibration['scale'], dtype=torch.float64)
        sd['hm.2.weight'] = (sd['hm.2.weight'].double() * sc.view(-1, 1, 1, 1)).float()
        sd['hm.2.bias'] = torch.tensor(calibration['bias'], dtype=torch.float64).float()
    else:
        sd = W.make_synthetic_state_dict(case['heads'], seed=case['seed'], hm_gain=case['hm_gain'])
    half = case['box_cells'] / 2
    if 'ltrb_amodal' in case['heads']:
        sd['ltrb_amodal.2.bias'] = torch.tensor([-half, -half, half, half])
    sd['wh.2.bias'] = torch.tensor([case['box_cells'], case['box_cells']])
    return sd


def e2e_mode_frames(case):
    """(images, meta) per frame: the scrolled N(0,1) stream of e2e_frames; [2,3,H,W] with the mirrored copy under
    flip_test (detector.py:225-226); ``pre_dets`` (first frame) / ``cur_dets`` in the meta for the public-detection case:
    a regular grid of provided detections every ``public_grid`` image px, in the shape
    tools/convert_mot_det_to_results.py:31-56 stores them"""
    from centertrack_amd.image import make_meta
    g = torch.Generator().manual_seed(case.get('frames_seed', case['seed'] + 7))      # (weights: case['seed'])
    T = case['T']
    base = torch.randn((3, case['H'], case['W'] + 4 * T), generator=g, dtype=torch.float64).float()
    meta = make_meta(case['H'], case['W'], case['orig_h'], case['orig_w'], down_ratio=4)
    pub = None
    if case.get('public_grid'):
        st = case['public_grid']
        pub = []
        for y in range(st // 2, case['orig_h'], st):
            for x in range(st // 2, case['orig_w'], st):
                bbox = [float(x - 15), float(y - 20), float(x + 15), float(y + 20)]
                pub.append({'bbox': bbox, 'score': 1.0, 'class': 1, 'ct': [float(x), float(y)]})
    for t in range(T):
        img = base[:, :, 4 * t:4 * t + case['W']].contiguous().unsqueeze(0)
        if case['opt'].get('flip_test'):
            img = torch.cat((img, torch.flip(img, [3])), 0)
        m = dict(meta)
        if pub is not None:
            m['cur_dets'] = [dict(d) for d in pub]
            if t == 0:
                m['pre_dets'] = [dict(d) for d in pub[::3]]
        yield img, m


def writer_case(seed=23):
    """Synthetic tracking results of 2 videos (4 + 3 frames) for the result-writer tests: image-id keyed dict of
    item lists like ``test.py`` collects (``results[img_id] = ret['results']``, test.py:99-110), the
    ``videos`` / ``video_to_images`` tables the dataset classes hold (generic_dataset.py:590-602), KITTI class
    names.  Items carry float32 numpy values like the post-process emits; some are inactive, ids are not
    contiguous, the 3D fields are present on a few."""
    rs = np.random.RandomState(seed)
    videos = [{'id': 1, 'file_name': 'MOT17-02-FRCNN'}, {'id': 2, 'file_name': '0004'}]
    video_to_images = {1: [], 2: []}
    results = {}
    img_id = 100
    for vid, nframes in ((1, 4), (2, 3)):
        ids = [7, 3, 12, 40]
        for fidx in range(nframes):
            img_id += 1
            video_to_images[vid].append({'id': img_id, 'frame_id': fidx + 1, 'video_id': vid})
            if vid == 1 and fidx == 2:
                continue                                   # an image without results is skipped by the writers
            items = []
            for j, tid in enumerate(ids):
                x0, y0 = rs.uniform(-20, 900), rs.uniform(-10, 500)
                bw, bh = rs.uniform(5, 200), rs.uniform(5, 300)
                it = {'score': np.float32(rs.uniform(0.3, 1.0)), 'class': int(rs.randint(1, 4)),
                      'ct': np.array([x0 + bw / 2, y0 + bh / 2], np.float32),
                      'tracking': rs.uniform(-3, 3, 2).astype(np.float32),
                      'bbox': np.array([x0, y0, x0 + bw, y0 + bh], np.float32),
                      'tracking_id': tid, 'age': 1, 'active': int(not (j == 1 and fidx == 1))}
                if vid == 2 and j % 2 == 0:
                    it.update({'dep': rs.uniform(3, 60, 1).astype(np.float32), 'alpha': float(rs.uniform(-3, 3)),
                               'dim': rs.uniform(-0.5, 4, 3).astype(np.float32),
                               'loc': rs.uniform(-30, 60, 3).astype(np.float32), 'rot_y': float(rs.uniform(-3, 3))})
                items.append(it)
            if fidx == 1:
                ids = ids[:-1] + [55]                      # a track ends, a new id appears
            results[img_id] = items
    return {'videos': videos, 'video_to_images': video_to_images, 'results': results,
            'kitti_class_name': ['Pedestrian', 'Car', 'Cyclist']}
