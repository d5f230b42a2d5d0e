"""Result writers: centertrack_amd/results_io.py writes byte-for-byte what the reference's
MOT.save_results / KITTITracking.save_results write (golden: tests/golden/writers.json, produced by running the
reference's own methods on scenarios.writer_case(), see tests/golden/make_golden.py gen_writers)."""
import json
import os

import numpy as np

from centertrack_amd import results_io
from scenarios import writer_case


def _read_tree(d):
    out = {}
    for root, _, names in os.walk(d):
        for n in names:
            with open(os.path.join(root, n)) as f:
                out[os.path.relpath(os.path.join(root, n), d)] = f.read()
    return out


def test_mot_and_kitti_files_equal_the_reference(golden_dir, tmp_path):
    with open(os.path.join(golden_dir, 'writers.json')) as f:
        gold = json.load(f)
    case = writer_case()
    d = str(tmp_path / 'mot')
    os.makedirs(d)
    results_io.save_mot_results(case['results'], d, case['videos'], case['video_to_images'], '17halfval')
    assert _read_tree(d) == gold['mot']
    d = str(tmp_path / 'kitti')
    os.makedirs(d)
    results_io.save_kitti_tracking_results(case['results'], d, case['videos'], case['video_to_images'],
                                           case['kitti_class_name'])
    assert _read_tree(d) == gold['kitti']
    # the inputs are not mutated (the reference's writer fills placeholders into the items in place)
    assert all('alpha' not in it for it in case['results'][102] if 'dep' not in it)


def test_native_structured_rows_write_the_same_lines():
    """the native host path returns numpy structured rows (fast_track.TRACK_DTYPE) instead of dicts"""
    from centertrack_amd.fast_track import TRACK_DTYPE
    case = writer_case()
    frames_d, frames_s = [], []
    for info in case['video_to_images'][1]:
        if info['id'] not in case['results']:
            continue
        items = case['results'][info['id']]
        arr = np.zeros(len(items), TRACK_DTYPE)
        for i, it in enumerate(items):
            for k in ('score', 'class', 'ct', 'tracking', 'bbox', 'tracking_id', 'age', 'active'):
                arr[i][k] = it[k]
        frames_d.append((info['frame_id'], items))
        frames_s.append((info['frame_id'], arr))
    assert results_io.mot_lines(frames_s) == results_io.mot_lines(frames_d)
    names = case['kitti_class_name']
    assert results_io.kitti_tracking_lines(frames_s, names) == results_io.kitti_tracking_lines(frames_d, names)


def test_mot_round_trip_and_edge_cases(tmp_path):
    assert results_io.mot_lines([]) == [] and results_io.kitti_tracking_lines([], ['a']) == []
    items = [{'bbox': np.array([10.004, 20.006, 30.0, 50.0], np.float32), 'tracking_id': 9, 'active': 1, 'score': 0.5,
              'class': 1},
             {'bbox': np.array([1, 2, 3, 4], np.float32), 'tracking_id': 2, 'active': 0, 'score': 0.9, 'class': 1}]
    lines = results_io.mot_lines([(5, items), (6, [])])
    assert lines == ['5,1,10.00,20.01,20.00,29.99,-1,-1,-1,-1']          # inactive dropped, ids renumbered from 1
    p = str(tmp_path / 'r.txt')
    with open(p, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    assert results_io.read_mot_results(p) == {5: [(1, 10.0, 20.01, 20.0, 29.99)]}


def test_public_detections_from_a_mot_det_file():
    """tools/convert_mot_det_to_results.py:31-56 per line: float32-parsed x,y,w,h -> x1y1x2y2 box, score 1, class 1"""
    lines = ['1,-1,100.5,20.25,30,60.5,0.93,-1,-1,-1', '1,-1,5,6,7,8,1,-1,-1,-1', '3,-1,0.1,0.2,0.3,0.4,0.5,-1,-1,-1', '']
    d = results_io.public_dets_from_mot(lines, frame_base=300)
    assert sorted(d) == [301, 303] and len(d[301]) == 2
    assert d[301][0] == {'bbox': [100.5, 20.25, 130.5, 80.75], 'score': 1.0, 'class': 1, 'ct': [115.5, 50.5]}
    x, y, w, h = (float(np.float32(v)) for v in (0.1, 0.2, 0.3, 0.4))
    assert d[303][0]['bbox'] == [x, y, x + w, y + h]
