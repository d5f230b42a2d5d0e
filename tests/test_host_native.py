"""CPU: the native (C++) host path -- post-process + score cut + greedy association + prior
heat-map blob parameters -- is identical to the reference-shaped Python path and to the
oracle on random multi-frame streams (IDs / age / active exact, values bitwise here)."""
import types

import numpy as np
import pytest

from centertrack_amd import _lib
from centertrack_amd import fast_track as FT
from centertrack_amd import image as IM
from centertrack_amd import ops
from centertrack_amd import post_process as PP
from centertrack_amd import tracker as TR
from centertrack_amd.detector import render_pre_hm
from oracle import post_process as OPP
from oracle import tracker as OTR


def _stream(rs, K, F, nobj=60):
    objs = [dict(p=rs.uniform(5, 120, 2), v=rs.normal(0, 1.5, 2), s=rs.uniform(2, 10)) for _ in range(nobj)]
    while True:
        rows = np.zeros((K, F), np.float32)
        rows[:, 0] = np.sort(rs.uniform(0.05, 1, K).astype(np.float32))[::-1]
        rows[:, 1] = rs.randint(0, 2, K)
        for j in range(K):
            o = objs[j % nobj]
            if j < nobj:
                o['p'] = o['p'] + o['v']
            c = np.floor(o['p'] + (0 if j < nobj else rs.normal(0, 20, 2)))
            rows[j, 2:4] = c
            rows[j, 4:8] = [c[0] - o['s'], c[1] - o['s'], c[0] + o['s'], c[1] + o['s']]
            rows[j, 8:12] = rows[j, 4:8]
            rows[j, 12:14] = -o['v'] + rs.normal(0, 0.3, 2)
        rows[:, 1:] = rows[rs.permutation(K), 1:]
        yield rows


@pytest.mark.parametrize('max_age', [-1, 2])
def test_native_host_path_equals_python_and_oracle(max_age):
    rs = np.random.RandomState(3)
    lay_list, F = ops.decode_layout(['reg', 'wh', 'tracking', 'ltrb_amodal'])
    lay = FT.row_layout(lay_list)
    K = 100
    meta = IM.make_meta(512, 512, 1080, 1920)
    trans = np.ascontiguousarray(IM.get_affine_transform(
        meta['c'], meta['s'], 0, (meta['out_width'], meta['out_height']), inv=1).astype(np.float32))
    opt = types.SimpleNamespace(out_thresh=0.4, new_thresh=0.4, max_age=max_age, hungarian=False, public_det=False)
    ft, pt, ot = FT.FastTracker(0.4, max_age, K), TR.Tracker(opt), OTR.Tracker(0.4, max_age)
    matched = 0
    for t, rows in zip(range(25), _stream(rs, K, F)):
        dec = {n: (rows[None, :, s] if n in ('scores', 'clses', 'xs', 'ys') else rows[None, :, s:s + w])
               for n, s, w in lay_list}
        dec['cts'] = rows[None, :, 2:4]
        got = ft.step(rows, lay, 0.4, trans).copy()
        res = PP.generic_post_process(opt, dec, [meta['c']], [meta['s']], meta['out_height'], meta['out_width'])[0]
        want = pt.step([r for r in res if r['score'] > 0.4])
        ores = OPP.generic_post_process(0.4, dec, [meta['c']], [meta['s']], meta['out_height'], meta['out_width'])[0]
        owant = ot.step([r for r in ores if r['score'] > 0.4])
        ids = [int(x['tracking_id']) for x in want]
        assert ids == [int(x['tracking_id']) for x in owant]
        assert [int(x) for x in got['tracking_id']] == ids, t
        assert [int(x) for x in got['active']] == [int(x['active']) for x in want]
        assert [int(x) for x in got['age']] == [int(x['age']) for x in want]
        assert [int(x) for x in got['class']] == [int(x['class']) for x in want]
        for w, g in zip(want, got):
            np.testing.assert_allclose(g['bbox'], np.asarray(w['bbox']), rtol=1e-6, atol=1e-4)
            np.testing.assert_allclose(g['ct'], np.asarray(w['ct']), rtol=1e-6, atol=1e-4)
            np.testing.assert_allclose(g['tracking'], np.asarray(w['tracking']), rtol=1e-5, atol=1e-4)
        matched += sum(int(x['active']) > 1 for x in want)
        # prior heat-map of the next frame: blob list rendered with numpy == python rendering
        n, prm = ft.prehm_params(0.5, meta['trans_input'], 512, 512)
        hm_py, _ = render_pre_hm(pt.tracks, meta, 0.5)
        out = np.zeros((512, 512), np.float32)
        ys, xs = np.mgrid[0:512, 0:512]
        for cx, cy, r in prm[:n]:
            sig = (2 * r + 1) / 6.0
            y0, y1, x0, x1 = max(0, cy - r), min(512, cy + r + 1), max(0, cx - r), min(512, cx + r + 1)
            gg = np.exp(-((xs[y0:y1, x0:x1] - cx) ** 2 + (ys[y0:y1, x0:x1] - cy) ** 2) / (2 * sig * sig))
            out[y0:y1, x0:x1] = np.maximum(out[y0:y1, x0:x1], gg.astype(np.float32))
        if t % 6 == 0:
            np.testing.assert_array_equal(out, hm_py)
    assert matched > 20, 'the synthetic stream should exercise association'
    assert ft.id_count == pt.id_count


def test_native_tracker_reset_and_empty_frames():
    lay = FT.row_layout(ops.decode_layout(['reg', 'wh', 'tracking'])[0])
    ft = FT.FastTracker(0.3, -1, 10)
    ident = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    rows = np.zeros((10, 10), np.float32)            # all scores 0 < out_thresh -> no detections
    assert len(ft.step(rows, lay, 0.3, ident)) == 0
    rows[0] = [0.9, 0, 5, 5, 3, 3, 7, 7, 0, 0]
    rows[1] = [0.3, 0, 9, 9, 8, 8, 10, 10, 0, 0]     # score == thresh: cut by the strict '>' of merge_outputs
    r = ft.step(rows, lay, 0.3, ident)
    assert len(r) == 1 and int(r['tracking_id'][0]) == 1 and ft.id_count == 1
    ft.reset()
    assert ft.id_count == 0 and len(ft.tracks) == 0


def test_python_post_process_matches_reference_golden(golden_dir):
    """centertrack_amd.post_process.generic_post_process (the Python host path: 3D fields, key points) against
    the reference's own outputs (tests/golden/post_process.json), every field bit-identical"""
    import json
    import os
    import types
    import scenarios as S
    g = json.load(open(os.path.join(golden_dir, 'post_process.json')))
    for case in S.postprocess_cases():
        opt = types.SimpleNamespace(out_thresh=case['out_thresh'])
        r = PP.generic_post_process(opt, {k: v.copy() for k, v in case['dets'].items()}, [case['c']], [case['s']],
                                    case['h'], case['w'], case['num_classes'], [case['calib']], case['height'],
                                    case['width'])[0]
        ref = g[case['name']]
        assert len(r) == len(ref)
        for a, b in zip(r, ref):
            assert sorted(a.keys()) == sorted(b.keys())
            for k in a:
                np.testing.assert_array_equal(np.asarray(a[k], np.float64), np.asarray(b[k], np.float64),
                                              err_msg='%s.%s' % (case['name'], k))


def test_python_tracker_matches_reference_golden(golden_dir):
    """centertrack_amd.tracker.Tracker (the Python host path that serves --hungarian, --public_det and pre_dets;
    SURVEY.md 8f rank 4) on every reference sequence of tests/golden/tracker.json: greedy, gating by size and
    class, max_age re-activation, Hungarian assignment, public-detection births -- ids / ages / active flags exact."""
    import json
    import os
    import scenarios as S
    g = json.load(open(os.path.join(golden_dir, 'tracker.json')))
    names = set()
    for seq in S.tracker_sequences():
        tr = TR.Tracker(types.SimpleNamespace(**seq['opt']))
        tr.init_track([dict(d) for d in seq.get('pre_dets', [])])
        for t, fr in enumerate(seq['frames']):
            dets = [{k: (np.array(v, np.float32) if isinstance(v, list) else v) for k, v in d.items()} for d in fr['dets']]
            ret = tr.step(dets, fr.get('public_det'))
            got = [{'tracking_id': int(x['tracking_id']), 'age': int(x['age']), 'active': int(x['active']),
                    'score': float(x['score']), 'class': int(x['class'])} for x in ret]
            assert got == g[seq['name']][t], (seq['name'], t)
        names.add(seq['name'])
    assert {'hungarian_cross', 'public_det', 'random_hungarian'} <= names


def test_native_tracker_matches_reference_golden_all_modes(golden_dir):
    """ct_tracker_step_dets / ct_tracker_init_tracks / ct_tracker_set_mode (native C++: greedy, Hungarian via
    ct_linear_assignment, public-detection births, max_age, pre_dets) on every reference sequence of
    tests/golden/tracker.json: ids / ages / active flags / order identical to the reference's Tracker"""
    import json
    import os
    import scenarios as S
    g = json.load(open(os.path.join(golden_dir, 'tracker.json')))
    for seq in S.tracker_sequences():
        o = seq['opt']
        tr = FT.FastTracker(o['new_thresh'], o['max_age'], 128, hungarian=o['hungarian'], public_det=o['public_det'])
        tr.init_tracks([dict(d) for d in seq.get('pre_dets', [])])
        for t, fr in enumerate(seq['frames']):
            ret = tr.step_dets([dict(d) for d in fr['dets']], fr.get('public_det'))
            got = [{'tracking_id': int(x['tracking_id']), 'age': int(x['age']), 'active': int(x['active']),
                    'score': float(x['score']), 'class': int(x['class'])} for x in ret]
            want = [dict(w, score=float(np.float32(w['score']))) for w in g[seq['name']][t]]
            assert got == want, (seq['name'], t)


def test_linear_assignment_equals_scipy_including_ties():
    """ct_linear_assignment == scipy.optimize.linear_sum_assignment pair for pair: random, heavily tied integer,
    and 1e18-gated matrices (the tracker's), tall / wide / empty shapes"""
    from scipy.optimize import linear_sum_assignment
    lib = _lib.load()
    rs = np.random.RandomState(0)
    for it in range(1500):
        nr, nc = rs.randint(0, 13), rs.randint(0, 13)
        kind = it % 4
        if kind == 0:
            c = rs.uniform(0, 100, (nr, nc))
        elif kind == 1:
            c = rs.randint(0, 4, (nr, nc)).astype(np.float64)
        elif kind == 2:
            c = rs.uniform(0, 100, (nr, nc))
            c[rs.uniform(size=(nr, nc)) < 0.6] = 1e18
        else:
            c = np.full((nr, nc), 1e18)
            m = rs.uniform(size=(nr, nc)) < 0.2
            c[m] = rs.uniform(0, 50, m.sum())
        c = np.ascontiguousarray(c, np.float64)
        r, cc = linear_sum_assignment(c)
        R, C = np.zeros(13, np.int32), np.zeros(13, np.int32)
        n = lib.ct_linear_assignment(c.ctypes.data, nr, nc, R.ctypes.data, C.ctypes.data)
        assert n == len(r) and np.array_equal(R[:n], r) and np.array_equal(C[:n], cc), (nr, nc, kind)
    bad = np.array([[1.0, np.nan]])
    assert lib.ct_linear_assignment(bad.ctypes.data, 1, 2, R.ctypes.data, C.ctypes.data) == -1


@pytest.mark.parametrize('hungarian,public_det,max_age', [(True, False, -1), (True, False, 3), (False, True, 2),
                                                          (True, True, 2), (False, False, 4)])
def test_native_tracker_equals_python_tracker_on_random_streams(hungarian, public_det, max_age):
    """40-frame random streams (objects appearing / disappearing, clutter, mixed classes, public detections near
    some objects): the native tracker and the reference-shaped Python tracker produce the same list frame by frame
    (ids, ages, active flags, order, boxes)"""
    rs = np.random.RandomState(17 + 2 * int(hungarian) + int(public_det) + max(max_age, 0))
    opt = types.SimpleNamespace(new_thresh=0.4, max_age=max_age, hungarian=hungarian, public_det=public_det)
    nat = FT.FastTracker(0.4, max_age, 128, hungarian=hungarian, public_det=public_det)
    pyt = TR.Tracker(opt)
    objs = [dict(p=rs.uniform(50, 900, 2), v=rs.normal(0, 6, 2), s=rs.uniform(15, 60), c=int(rs.randint(1, 3)))
            for _ in range(14)]
    births = 0
    for t in range(40):
        dets = []
        for o in objs:
            o['p'] = o['p'] + o['v']
            if rs.uniform() < 0.85:                        # detected this frame
                ct = (o['p'] + rs.normal(0, 1.5, 2)).astype(np.float32)
                dets.append({'score': float(np.float32(rs.uniform(0.3, 1.0))), 'class': o['c'], 'ct': ct,
                             'tracking': (-o['v'] + rs.normal(0, 1.0, 2)).astype(np.float32),
                             'bbox': np.array([ct[0] - o['s'], ct[1] - o['s'], ct[0] + o['s'], ct[1] + o['s']], np.float32)})
        for _ in range(rs.randint(0, 5)):                  # clutter
            ct = rs.uniform(0, 1000, 2).astype(np.float32)
            dets.append({'score': float(np.float32(rs.uniform(0.3, 0.9))), 'class': int(rs.randint(1, 3)), 'ct': ct,
                         'tracking': rs.normal(0, 3, 2).astype(np.float32),
                         'bbox': np.array([ct[0] - 20, ct[1] - 20, ct[0] + 20, ct[1] + 20], np.float32)})
        dets.sort(key=lambda d: -d['score'])
        pub = [{'ct': (o['p'] + rs.normal(0, 4, 2)).tolist()} for o in objs if rs.uniform() < 0.6] if public_det else None
        if t % 13 == 12:
            objs[rs.randint(len(objs))] = dict(p=rs.uniform(50, 900, 2), v=rs.normal(0, 6, 2), s=rs.uniform(15, 60),
                                               c=int(rs.randint(1, 3)))
        want = pyt.step([dict(d) for d in dets], pub)
        got = nat.step_dets(dets, pub)
        assert [int(x['tracking_id']) for x in got] == [int(x['tracking_id']) for x in want], t
        assert [int(x['age']) for x in got] == [int(x['age']) for x in want]
        assert [int(x['active']) for x in got] == [int(x['active']) for x in want]
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a['bbox'], np.asarray(b['bbox'], np.float32))
        births = max(births, nat.id_count)
    assert births > 14 and nat.id_count == pyt.id_count


def test_native_results_as_dicts_carry_the_3d_fields_like_the_reference():
    """ddd heads (nuScenes): the native rows + fast_track.as_dicts(dets, calib) give the dicts the reference's
    generic_post_process + Tracker give -- dep, dim, alpha, loc, rot_y, amodal ct, also on tracks carried over
    without a detection (max_age) -- so Detector.run()['results'] is complete on the native host path too."""
    rs = np.random.RandomState(9)
    names = ['reg', 'wh', 'tracking', 'dep', 'rot', 'dim', 'amodel_offset']
    lay_list, F = ops.decode_layout(names)
    lay = FT.row_layout(lay_list)
    K = 40
    meta = IM.make_meta(448, 800, 900, 1600)
    calib = meta['calib']
    trans = np.ascontiguousarray(IM.get_affine_transform(
        meta['c'], meta['s'], 0, (meta['out_width'], meta['out_height']), inv=1).astype(np.float32))
    opt = types.SimpleNamespace(out_thresh=0.3, new_thresh=0.3, max_age=2, hungarian=False, public_det=False)
    ft, pt = FT.FastTracker(0.3, 2, K), TR.Tracker(opt)
    off = {n: s for n, s, _ in lay_list}
    carried = {}
    objs = [dict(p=rs.uniform(20, 180, 2), v=rs.normal(0, 1.0, 2)) for _ in range(10)]
    seen_carried = 0
    for t in range(8):
        rows = np.zeros((K, F), np.float32)
        rows[:, 0] = np.sort(rs.uniform(0.05, 1, K).astype(np.float32))[::-1]
        rows[:, 1] = rs.randint(0, 3, K)
        for j in range(K):
            o = objs[j % len(objs)]
            c = np.floor(o['p'] + (0 if j < len(objs) else rs.normal(0, 30, 2)))
            rows[j, 2:4] = c
            half = rs.uniform(3, 9)
            rows[j, off['bboxes']:off['bboxes'] + 4] = [c[0] - half, c[1] - half, c[0] + half, c[1] + half]
            rows[j, off['tracking']:off['tracking'] + 2] = -o['v'] + rs.normal(0, 0.2, 2)
            rows[j, off['dep']] = rs.uniform(3, 60)
            rows[j, off['rot']:off['rot'] + 8] = rs.normal(0, 1, 8)
            rows[j, off['dim']:off['dim'] + 3] = rs.uniform(0.5, 4, 3)
            rows[j, off['amodel_offset']:off['amodel_offset'] + 2] = rs.normal(0, 1, 2)
        if t in (3, 4):                                  # the strongest object goes undetected: its track is carried
            rows[0, 0] = 0.01
            rows[:, :] = rows[np.argsort(-rows[:, 0], kind='stable')]
        for o in objs:
            o['p'] = o['p'] + o['v']
        dec = {n: (rows[None, :, s] if n in ('scores', 'clses', 'xs', 'ys') else rows[None, :, s:s + w])
               for n, s, w in lay_list}
        dec['cts'] = rows[None, :, 2:4]
        got = FT.as_dicts(ft.step(rows, lay, 0.3, trans).copy(), dec, 0, calib, carried)
        res = PP.generic_post_process(opt, {k: v.copy() for k, v in dec.items()}, [meta['c']], [meta['s']],
                                      meta['out_height'], meta['out_width'], 3, [calib], 900, 1600)[0]
        want = pt.step([r for r in res if r['score'] > 0.3])
        assert [r['tracking_id'] for r in got] == [r['tracking_id'] for r in want], t
        for a, b in zip(got, want):
            assert a['age'] == b['age'] and a['active'] == b['active']
            for k in ('dep', 'dim', 'alpha', 'loc', 'rot_y', 'ct', 'bbox'):
                np.testing.assert_allclose(np.asarray(a[k], np.float64), np.asarray(b[k], np.float64), rtol=1e-6,
                                           atol=1e-5, err_msg='frame %d %s' % (t, k))
            seen_carried += int(a['active'] == 0)
    assert seen_carried > 0


def test_result_dicts_do_not_alias_the_reused_rows_buffer():
    """the packed rows live in ONE pinned buffer that every step overwrites; dicts handed out for
    frame t (dep / dim / nuscenes_att / velocity and the tracks carried by max_age) must not change when the buffer
    is rewritten for frame t+1 -- the reference's per-frame ``.cpu().numpy()`` arrays never do."""
    rs = np.random.RandomState(2)
    names = ['reg', 'wh', 'tracking', 'dep', 'rot', 'dim', 'amodel_offset', 'nuscenes_att', 'velocity']
    lay_list, F = ops.decode_layout(names)
    lay = FT.row_layout(lay_list)
    off = {n: s for n, s, _ in lay_list}
    K = 8
    meta = IM.make_meta(448, 800, 900, 1600)
    trans = np.ascontiguousarray(IM.get_affine_transform(
        meta['c'], meta['s'], 0, (meta['out_width'], meta['out_height']), inv=1).astype(np.float32))
    rows = np.zeros((1, K, F), np.float32)                # the "pinned buffer"
    ft = FT.FastTracker(0.3, 2, K)
    carried = {}

    def fill(seed):
        r = np.random.RandomState(seed)
        rows[0] = r.uniform(0.5, 4, (K, F)).astype(np.float32)
        rows[0, :, 0] = np.linspace(0.9, 0.4, K)
        rows[0, :, 1] = 0
        for j in range(K):
            c = np.array([20.0 + 20 * j, 50.0])
            rows[0, j, 2:4] = c
            rows[0, j, off['bboxes']:off['bboxes'] + 4] = [c[0] - 6, c[1] - 6, c[0] + 6, c[1] + 6]
            rows[0, j, off['tracking']:off['tracking'] + 2] = 0

    def unpack(buf):
        d = {n: (buf[..., s] if n in ('scores', 'clses', 'xs', 'ys') else buf[..., s:s + w]) for n, s, w in lay_list}
        d['cts'] = buf[..., 2:4]
        return d

    fill(1)
    frame0 = FT.as_dicts(ft.step(rows[0], lay, 0.3, trans).copy(), unpack(rows), 0, meta['calib'], carried)
    snap = [{k: np.array(v, copy=True) for k, v in d.items()} for d in frame0]
    fill(2)                                               # the next step's D2H lands in the same memory
    rows[0, :, 0] = 0.01                                  # ... and detects nothing: every track is carried over
    frame1 = FT.as_dicts(ft.step(rows[0], lay, 0.3, trans).copy(), unpack(rows), 0, meta['calib'], carried)
    for d, s in zip(frame0, snap):
        for k in s:
            np.testing.assert_array_equal(np.asarray(d[k]), s[k], err_msg=k)
    assert len(frame1) == len(frame0) and all(r['active'] == 0 for r in frame1)
    for d, s in zip(frame1, snap):                        # carried tracks show the fields of their last detection
        for k in ('dep', 'dim', 'nuscenes_att', 'velocity'):
            np.testing.assert_array_equal(np.asarray(d[k]), s[k], err_msg=k)


def test_native_tracker_grows_past_its_initial_capacity():
    """with max_age > 0 unmatched tracks accumulate beyond 2K+64; the step completes, the result buffer
    grows, and ids / order stay those of the Python tracker."""
    K = 6
    lay_list, F = ops.decode_layout(['reg', 'wh', 'tracking'])
    lay = FT.row_layout(lay_list)
    meta = IM.make_meta(512, 512, 512, 512)
    trans = np.ascontiguousarray(IM.get_affine_transform(
        meta['c'], meta['s'], 0, (meta['out_width'], meta['out_height']), inv=1).astype(np.float32))
    opt = types.SimpleNamespace(out_thresh=0.3, new_thresh=0.3, max_age=40, hungarian=False, public_det=False)
    ft, pt = FT.FastTracker(0.3, 40, K), TR.Tracker(opt)
    cap0 = ft.cap
    for t in range(30):                                   # K brand-new far-apart objects per frame, none re-detected
        rows = np.zeros((K, F), np.float32)
        rows[:, 0] = np.linspace(0.9, 0.5, K)
        for j in range(K):
            c = np.array([3.0 + 7 * ((t * K + j) % 17), 3.0 + 7 * ((t * K + j) // 17)], np.float32)
            rows[j, 2:4] = c
            rows[j, 4:8] = [c[0] - 1, c[1] - 1, c[0] + 1, c[1] + 1]
        dec = {n: (rows[None, :, s] if n in ('scores', 'clses', 'xs', 'ys') else rows[None, :, s:s + w])
               for n, s, w in lay_list}
        dec['cts'] = rows[None, :, 2:4]
        got = ft.step(rows, lay, 0.3, trans).copy()
        res = PP.generic_post_process(opt, dec, [meta['c']], [meta['s']], meta['out_height'], meta['out_width'])[0]
        want = pt.step([r for r in res if r['score'] > 0.3])
        assert [int(x['tracking_id']) for x in got] == [int(x['tracking_id']) for x in want], t
        assert [int(x['age']) for x in got] == [int(x['age']) for x in want], t
    assert len(got) > cap0 and ft.cap > cap0
    n, prm = ft.prehm_params(0.3, meta['trans_input'], 512, 512)
    assert n == K                                         # only this frame's detections are active


def test_native_key_points_equal_the_python_post_process():
    """pose task: the native rows + as_dicts(trans_inv) give the key points `hps` of generic_post_process
    (post_process.py:51-54) bit for bit -- ct_transform_points is the float32 [2,3] affine of the boxes"""
    rs = np.random.RandomState(12)
    lay_list, F0 = ops.decode_layout(['reg', 'wh', 'tracking'])
    J = 17
    F = F0 + 2 * J + 1
    lay = FT.row_layout(lay_list)
    K = 30
    meta = IM.make_meta(512, 512, 480, 640)
    trans = np.ascontiguousarray(IM.get_affine_transform(
        meta['c'], meta['s'], 0, (meta['out_width'], meta['out_height']), inv=1).astype(np.float32))
    opt = types.SimpleNamespace(out_thresh=0.3, new_thresh=0.3, max_age=-1, hungarian=False, public_det=False)
    rows = np.zeros((K, F), np.float32)
    rows[:, 0] = np.sort(rs.uniform(0.05, 1, K).astype(np.float32))[::-1]
    rows[:, 2:4] = np.floor(rs.uniform(5, 120, (K, 2)))
    rows[:, 4:8] = np.concatenate((rows[:, 2:4] - 4, rows[:, 2:4] + 4), 1)
    rows[:, F0:F0 + 2 * J] = rs.uniform(-5, 130, (K, 2 * J)).astype(np.float32)
    dec = {n: (rows[None, :, s] if n in ('scores', 'clses', 'xs', 'ys') else rows[None, :, s:s + w]) for n, s, w in lay_list}
    dec['cts'] = rows[None, :, 2:4]
    dec['hps'] = rows[None, :, F0:F0 + 2 * J]
    ft = FT.FastTracker(0.3, -1, K)
    got = FT.as_dicts(ft.step(rows, lay, 0.3, trans).copy(), dec, 0, None, None, trans)
    want = PP.generic_post_process(opt, dec, [meta['c']], [meta['s']], meta['out_height'], meta['out_width'])[0]
    want = [r for r in want if r['score'] > 0.3]
    assert len(got) == len(want) > 5
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a['hps'], b['hps'])
        np.testing.assert_array_equal(a['bbox'], b['bbox'])
