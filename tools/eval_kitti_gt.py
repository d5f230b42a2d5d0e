#!/usr/bin/env python
"""Recipe: hot path's tracks -> on-disk KITTI tracking files -> the REFERENCE's evaluator.

    python tools/eval_kitti_gt.py [mode ...] [--split val_half|full] [--out DIR] [--write-golden]

Build container only (the evaluator and its ground truth live in /root/reference).  For each mode of
tests/golden/kitti_gt_tracks.npz (tests/golden/make_kitti_gt.py: detections derived from the reference's KITTI
val_half ground truth) the detections are replayed through the NATIVE tracker (``ct_tracker_step_dets``), collected
as ``{image_id: [items]}`` the way test.py:109 collects ``Detector.run(...)['results']``, written by
``centertrack_amd.results_io.save_kitti_tracking_results`` (= ``KITTITracking.save_results``,
datasets/kitti_tracking.py:51-97) and scored by ``src/tools/eval_kitti_track/evaluate_tracking.py`` run UNMODIFIED
as a subprocess in ``<reference>/src`` exactly like ``KITTITracking.run_eval`` does (kitti_tracking.py:99-102:
``python tools/eval_kitti_track/evaluate_tracking.py <dir>/results_kitti_tracking/ val_half``).  The evaluator's
summary files are parsed into a dict; ``--write-golden`` stores them as tests/golden/kitti_gt_eval.json.
``--split full``: the 21 complete training videos (``label_02``, kitti_gt_tracks_full.npz), scored the way the evaluator
scores a run without a split argument (evaluate_tracking.py:975: ``split_version = ''`` -> ``evaluate_tracking.seqmap``,
``label_02``) -> tests/golden/kitti_gt_eval_full.json."""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))

KEYS = {  # summary line -> key (first occurrence; the file repeats "ID-switches" / "Missed Targets" as ratios)
    'Multiple Object Tracking Accuracy (MOTA)': 'MOTA', 'Multiple Object Tracking Precision (MOTP)': 'MOTP',
    'Multiple Object Tracking Accuracy (MOTAL)': 'MOTAL', 'Multiple Object Detection Accuracy (MODA)': 'MODA',
    'Multiple Object Detection Precision (MODP)': 'MODP', 'Recall': 'recall', 'Precision': 'precision', 'F1': 'F1',
    'False Alarm Rate': 'false alarm rate', 'Mostly Tracked': 'mostly tracked', 'Partly Tracked': 'partly tracked',
    'Mostly Lost': 'mostly lost', 'True Positives': 'true positives', 'Ignored True Positives': 'ignored true positives',
    'False Positives': 'false positives', 'False Negatives': 'missed', 'ID-switches': 'id-switches',
    'Fragmentations': 'fragmentations', 'Ground Truth Objects (Total)': 'gt objects',
    'Ground Truth Trajectories': 'gt trajectories', 'Tracker Objects (Total)': 'tracker objects',
    'Tracker Trajectories': 'tracker trajectories'}


def parse_summary(path):
    out = {}
    with open(path) as f:
        for line in f:
            p = line.rsplit(None, 1)
            if len(p) == 2 and p[0].strip() in KEYS and KEYS[p[0].strip()] not in out:
                v = float(p[1])
                out[KEYS[p[0].strip()]] = int(v) if v == int(v) and '.' not in p[1] else v
    return out


def track_mode(gold, mode):
    """replay one mode through the native tracker -> (results {image_id: structured rows}, videos, video_to_images)"""
    import make_kitti_gt as G
    from centertrack_amd import fast_track as FT
    name, kind, hung, public, max_age = [m for m in G.MODES if m[0] == mode][0]
    dets, ptr = gold[kind + '.dets'], gold[kind + '.ptr']
    pub = gold.get(kind + '.public', np.zeros((0, 2), np.float32))
    pp = gold.get(kind + '.public_ptr', np.zeros(len(ptr), np.int32))
    results, videos, v2i = {}, [], {}
    f = 0
    for vid, (seq, n) in enumerate(zip(gold['seq_names'], gold['seq_frames']), 1):
        videos.append({'id': vid, 'file_name': '%04d' % seq})
        v2i[vid] = []
        ft = FT.FastTracker(G.NEW_THRESH[kind], max_age, 100, hungarian=bool(hung), public_det=bool(public))
        for t in range(n):
            items = G.frame_items(dets[ptr[f]:ptr[f + 1]])
            pitems = G.public_items(pub[pp[f]:pp[f + 1]]) if public else None
            if public and t == 0:
                ft.init_tracks([{'score': 1.0, 'class': 1, 'ct': p['ct'], 'bbox': [p['ct'][0] - 20, p['ct'][1] - 20,
                                                                                 p['ct'][0] + 20, p['ct'][1] + 20]}
                                for p in pitems])
            img_id = f + 1
            results[img_id] = ft.step_dets(FT.items_to_array(items), pitems).copy()
            v2i[vid].append({'id': img_id, 'frame_id': t + 1})      # 1-based frame ids; the writer stores frame_id - 1
            f += 1
    return results, videos, v2i


def evaluate(modes, out_dir, ref='/root/reference', split='val_half'):
    import make_kitti_gt as G
    from centertrack_amd import results_io
    gold = dict(np.load(os.path.join(REPO, 'tests', 'golden', G.SPLITS[split][2])))
    res = {}
    for mode in modes:
        d = os.path.join(out_dir, mode)
        os.makedirs(d, exist_ok=True)
        results, videos, v2i = track_mode(gold, mode)
        rdir = results_io.save_kitti_tracking_results(results, d, videos, v2i)
        p = subprocess.run([sys.executable, 'tools/eval_kitti_track/evaluate_tracking.py', rdir + '/'] + ([split] if split != 'full' else []),
                           cwd=os.path.join(ref, 'src'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))      # (nothing is written into the reference tree)
        if p.returncode != 0 or 'Thank you for participating' not in p.stdout:
            raise RuntimeError('reference evaluator failed on %s:\n%s' % (rdir, p.stdout[-2000:]))
        res[mode] = {c: parse_summary(os.path.join(rdir, 'summary_%s.txt' % c)) for c in ('car', 'pedestrian')}
    return res


def main():
    import tempfile
    import make_kitti_gt as G
    split = 'val_half'
    argv = list(sys.argv[1:])
    if '--split' in argv:
        split = argv[argv.index('--split') + 1]
        argv.remove(split)
    args = [a for a in argv if not a.startswith('--')]
    modes = args or [m[0] for m in G.MODES]
    out = None
    if '--out' in sys.argv:
        out = sys.argv[sys.argv.index('--out') + 1]
        modes = [m for m in modes if m != out]
    with tempfile.TemporaryDirectory() as tmp:
        res = evaluate(modes, out or tmp, split=split)
    for mode, r in res.items():
        for c, s in r.items():
            print('%-30s %-10s MOTA %.4f MOTP %.4f recall %.4f precision %.4f FP %d FN %d IDs %d frag %d' % (
                mode, c, s['MOTA'], s['MOTP'], s['recall'], s['precision'], s['false positives'], s['missed'],
                s['id-switches'], s['fragmentations']))
    if '--write-golden' in sys.argv:
        name = 'kitti_gt_eval.json' if split == 'val_half' else 'kitti_gt_eval_%s.json' % split
        with open(os.path.join(REPO, 'tests', 'golden', name), 'w') as f:
            json.dump(res, f, indent=1, sort_keys=True)
        print('tests/golden/%s written' % name)


if __name__ == '__main__':
    main()
