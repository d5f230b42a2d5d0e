#!/usr/bin/env python
"""Real-trajectory golden streams for the association step, made from the reference's OWN data and code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_kitti_gt.py [--split val_half|full] [--check]

Input: the KITTI tracking ground truth the reference ships for its evaluator,
``src/tools/eval_kitti_track/data/tracking/label_02_val_half/*.txt`` (21 sequences, 4 030 frames, ~30 k boxes of
real trajectories; the split its KITTI experiments are scored on, experiments/kitti_half.sh) and -- ``--split full`` ->
``kitti_gt_tracks_full.npz`` -- ``label_02/*.txt``, the 21 COMPLETE training videos (8 029 frames by the evaluator's
count, ~65 k label rows: the train halves are their first halves, so this is all the real tracking data in the
checkout).  Every frame's GT boxes are turned into
the detections ``Detector.run`` would hand to ``Tracker.step`` (utils/tracker.py:28): ``score``, ``class``
(datasets/kitti_tracking.py:19: Pedestrian 1, Car 2, Cyclist 3), ``bbox``, ``ct`` = box centre, ``tracking`` =
previous-frame GT centre - current centre (the displacement head's target, generic_dataset.py:409-411), all
float32 like ``generic_post_process`` produces them.  Two detection sets:

  clean   every Pedestrian / Car / Cyclist box, score 1, exact displacement (0 for an object without a
          previous-frame box)
  noisy   seeded (RandomState(317), the reference's default seed opts.py:48): score U(0.40, 1), box jitter
          N(0, 1.5 px), displacement jitter N(0, 2 px), 8 % drop-outs, Van -> Car and Person -> Pedestrian
          impostors, Poisson(0.4) false positives per frame, rows in score-descending order like the decode
          emits them -- so births below new_thresh, deaths, re-births, max_age carry-over and the size / class
          gates all fire; and a public-detection set (GT boxes, 15 % dropped, N(0, 2 px) jitter) for
          --public_det

Output of the REFERENCE's ``Tracker`` (imported unmodified through ref_import; its removed sklearn helper is
scipy's linear_sum_assignment) in every mode -- greedy / --hungarian, max_age 0 / 2, private /
--public_det with ``init_track(pre_dets)`` on a video's first frame (test.py:88-107, detector.py:101-103) --
as (tracking_id, age, active, source detection) per returned track, in the returned order.  Stored in
``tests/golden/kitti_gt_tracks.npz``; ``tests/test_kitti_gt_tracks.py`` replays the detections through
``ct_tracker_step_dets`` (native C++), ``centertrack_amd/tracker.py`` and the oracle: everything must be
IDENTICAL over all 4 030 frames.  ``tools/eval_kitti_gt.py`` then writes the tracks with
``results_io.save_kitti_tracking_results`` and runs the reference's evaluator on the files.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

CLASS_ID = {'Pedestrian': 1, 'Car': 2, 'Cyclist': 3}          # datasets/kitti_tracking.py:19
IMPOSTOR = {'Van': 2, 'Person': 1}                            # neighbouring classes of the evaluator (evaluate_tracking.py:247-252)
NEW_THRESH = {'clean': 0.4, 'noisy': 0.5}                     # experiments/kitti_half.sh:5 track_thresh 0.4; noisy: births gated
MODES = [  # name, detection set, hungarian, public_det, max_age
    ('clean_greedy', 'clean', 0, 0, -1),
    ('clean_hungarian', 'clean', 1, 0, -1),
    ('noisy_greedy', 'noisy', 0, 0, -1),
    ('noisy_greedy_age2', 'noisy', 0, 0, 2),
    ('noisy_hungarian', 'noisy', 1, 0, -1),
    ('noisy_hungarian_age2', 'noisy', 1, 0, 2),
    ('noisy_public', 'noisy', 0, 1, -1),
    ('noisy_public_hungarian_age2', 'noisy', 1, 1, 2),
]
DET_COLS = ('score', 'class', 'ct_x', 'ct_y', 'tracking_x', 'tracking_y', 'x1', 'y1', 'x2', 'y2', 'gt_id')


def label_dir(ref):
    return os.path.join(ref, 'src', 'tools', 'eval_kitti_track', 'data', 'tracking')


SPLITS = {  # name -> (seqmap, label directory, fixture): evaluate_tracking.py:99-100, 120-124
    'val_half': ('evaluate_trackingval_half.seqmap', 'label_02_val_half', 'kitti_gt_tracks.npz'),
    'full': ('evaluate_tracking.seqmap', 'label_02', 'kitti_gt_tracks_full.npz'),
}


def read_sequences(ref, split='val_half'):
    """[(name, n_frames, {frame: [(gt_id, type, x1, y1, x2, y2)]})] in seqmap order (evaluate_tracking.py:103-108)"""
    root = label_dir(ref)
    seqs = []
    seqmap, labels, _ = SPLITS[split]
    with open(os.path.join(root, seqmap)) as f:
        for line in f:
            p = line.split(' ')
            if len(p) < 4:
                continue
            name, n = '%04d' % int(p[0]), int(p[3]) - int(p[2]) + 1
            frames = {}
            with open(os.path.join(root, labels, name + '.txt')) as g:
                for row in g:
                    q = row.split(' ')
                    frames.setdefault(int(q[0]), []).append(
                        (int(q[1]), q[2], float(q[6]), float(q[7]), float(q[8]), float(q[9])))
            seqs.append((name, n, frames))
    return seqs


def synth_detections(seqs, kind, seed=317):
    """-> (dets float32 [n, 11] in DET_COLS order, frame_ptr int32 [n_frames_total + 1], seq_frames int32 [n_seq],
    public float32 [m, 2], public_ptr int32)"""
    rs = np.random.RandomState(seed)
    rows, ptr, pub_rows, pub_ptr, seq_frames = [], [0], [], [0], []
    for name, n, frames in seqs:
        seq_frames.append(n)
        prev = {}
        for t in range(n):
            cur, out, pub = {}, [], []
            for gid, typ, x1, y1, x2, y2 in frames.get(t, []):
                if typ in CLASS_ID:
                    cls = CLASS_ID[typ]
                elif kind == 'noisy' and typ in IMPOSTOR:
                    cls = IMPOSTOR[typ]
                else:
                    continue
                c = np.array([(x1 + x2) / 2, (y1 + y2) / 2], np.float64)
                cur[gid] = c
                disp = (prev[gid] - c) if gid in prev else np.zeros(2)
                box = np.array([x1, y1, x2, y2], np.float64)
                score = 1.0
                if kind == 'noisy':
                    keep = rs.uniform() >= 0.08
                    score = rs.uniform(0.40, 1.0)
                    box = box + rs.normal(0, 1.5, 4)
                    disp = disp + rs.normal(0, 2.0, 2)
                    if rs.uniform() >= 0.15 and typ in CLASS_ID:
                        pub.append(c + rs.normal(0, 2.0, 2))
                    if not keep:
                        continue
                    if box[2] < box[0] + 1 or box[3] < box[1] + 1:
                        continue
                ct = np.array([(box[0] + box[2]) / 2, (box[1] + box[3]) / 2])
                out.append([score, cls, ct[0], ct[1], disp[0], disp[1], box[0], box[1], box[2], box[3], gid])
            if kind == 'noisy':
                for _ in range(rs.poisson(0.4)):
                    w, h = rs.uniform(15, 120), rs.uniform(15, 120)
                    cx, cy = rs.uniform(0, 1242), rs.uniform(100, 375)
                    out.append([rs.uniform(0.40, 0.8), rs.randint(1, 4), cx, cy, rs.normal(0, 3), rs.normal(0, 3),
                                cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, -1])
                out.sort(key=lambda r: -np.float32(r[0]))          # stable: the decode's score-descending order
            rows += out
            ptr.append(len(rows))
            pub_rows += [list(p) for p in pub]
            pub_ptr.append(len(pub_rows))
            prev = cur
    return (np.array(rows, np.float32).reshape(-1, len(DET_COLS)), np.array(ptr, np.int32),
            np.array(seq_frames, np.int32), np.array(pub_rows, np.float32).reshape(-1, 2), np.array(pub_ptr, np.int32))


def frame_items(dets):
    """rows of one frame -> the dicts generic_post_process hands to the tracker (post_process.py:36-50: np.float32
    scalars and arrays)"""
    return [{'score': r[0], 'class': int(r[1]), 'ct': r[2:4].copy(), 'tracking': r[4:6].copy(),
             'bbox': r[6:10].copy(), 'src': i} for i, r in enumerate(dets)]


def public_items(pub):
    return [{'ct': p.copy()} for p in pub]


def run_tracker(make_tracker, dets, ptr, seq_frames, pub, pub_ptr, public_det):
    """one tracker per video (detector.reset_tracking, test.py:97); returns int16 [m, 4] rows
    (tracking_id, age, active, source detection of the frame or -1 for a carried track) + frame pointers"""
    out, optr = [], [0]
    f = 0
    for n in seq_frames:
        tr = make_tracker()
        for t in range(n):
            items = frame_items(dets[ptr[f]:ptr[f + 1]])
            pitems = public_items(pub[pub_ptr[f]:pub_ptr[f + 1]]) if public_det else None
            if public_det and t == 0:
                # detector.py:101-103: the first frame of a video starts tracks from the provided detections
                tr.init_track([{'score': 1.0, 'class': 1, 'ct': p['ct'].copy(),
                                'bbox': np.array([p['ct'][0] - 20, p['ct'][1] - 20, p['ct'][0] + 20, p['ct'][1] + 20],
                                                 np.float32), 'src': -1} for p in pitems])
            ret = tr.step(items, pitems)
            out += [[int(r['tracking_id']), int(r['age']), int(r['active']),
                     int(r['src']) if int(r['age']) == 1 else -1] for r in ret]
            optr.append(len(out))
            f += 1
    return np.array(out, np.int16).reshape(-1, 4), np.array(optr, np.int32)


def main():
    import ref_import
    ref = ref_import.install()
    import types
    from utils.tracker import Tracker          # the reference's own class
    split = sys.argv[sys.argv.index('--split') + 1] if '--split' in sys.argv else 'val_half'
    fixture = os.path.join(HERE, SPLITS[split][2])
    seqs = read_sequences(ref, split)
    out = {'seq_names': np.array([int(s[0]) for s in seqs], np.int32)}
    sets = {}
    for kind in ('clean', 'noisy'):
        d, p, sf, pub, pp = synth_detections(seqs, kind)
        sets[kind] = (d, p, sf, pub, pp)
        out[kind + '.dets'], out[kind + '.ptr'], out['seq_frames'] = d, p, sf
        if kind == 'noisy':
            out[kind + '.public'], out[kind + '.public_ptr'] = pub, pp
        print(kind, 'detections', d.shape, 'frames', len(p) - 1, 'public', pub.shape)
    for name, kind, hung, public, max_age in MODES:
        d, p, sf, pub, pp = sets[kind]
        opt = types.SimpleNamespace(new_thresh=NEW_THRESH[kind], max_age=max_age, hungarian=bool(hung),
                                    public_det=bool(public))
        rows, optr = run_tracker(lambda: Tracker(opt), d, p, sf, pub, pp, public)
        out[name + '.tracks'], out[name + '.ptr'] = rows, optr
        print(name, 'tracks', rows.shape, 'ids', int(rows[:, 0].max()), 'carried', int((rows[:, 3] < 0).sum()))
    if '--check' in sys.argv:          # re-run of the reference against the committed fixture, nothing written
        have = np.load(fixture)
        bad = [k for k in out if k not in have or not np.array_equal(have[k], out[k])] + \
              [k for k in have.files if k not in out]
        print('CHECK', 'FAILED ' + ','.join(bad) if bad else 'OK: %d arrays identical' % len(out))
        sys.exit(1 if bad else 0)
    np.savez_compressed(fixture, **out)
    print(os.path.basename(fixture), os.path.getsize(fixture), 'bytes')


if __name__ == '__main__':
    main()
