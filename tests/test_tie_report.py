"""tools/tie_report.py's comparator on hand-made streams (CPU): identical streams, a rank swap inside a tie, a threshold
flip that breaks the ids, an unexplained divergence."""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools'))
import tie_report as TR  # noqa: E402


def _frame(scores, ids):
    n = len(scores)
    d = {'scores': np.array(scores, np.float32), 'clses': np.zeros(n, np.float32), 'xs': np.arange(n, dtype=np.float32),
         'ys': np.zeros(n, np.float32), 'bboxes': np.stack([np.arange(n) * 10.0, np.zeros(n), np.arange(n) * 10.0 + 8, np.full(n, 8.0)], 1).astype(np.float32),
         'tracking': np.zeros((n, 2), np.float32)}
    res = [{'score': float(s), 'class': 1, 'id': i, 'bbox': [float(v) * 8 for v in d['bboxes'][k]], 'active': 1}
           for k, (s, i) in enumerate(zip(scores, ids)) if s > 0.4]
    return d, res


def test_identical_streams_report_nothing():
    ref = [_frame([0.9, 0.7, 0.5, 0.2], [1, 2, 3, 0]), _frame([0.8, 0.6, 0.45, 0.1], [1, 2, 3, 0])]
    acc = TR.Acc()
    TR.compare_stream('x', copy.deepcopy(ref), ref, 0.4, [0.4, 0.5], (acc,))
    r = acc.report()
    assert r['frames_compared'] == 2 and r['detections_compared'] == 6
    assert r['abs_dscore']['max'] == 0 and r['rank_swaps']['count'] == 0 and r['threshold_flips']['count'] == 0
    assert r['streams_with_id_divergence'] == 0 and not r['unexplained_divergences']


def test_rank_swap_and_birth_tie_permutation():
    ref = [_frame([0.9, 0.700001, 0.7, 0.2], [1, 2, 3, 0])]
    d, res = _frame([0.9, 0.700001, 0.7, 0.2], [1, 3, 2, 0])
    # ours ranks the two tied detections the other way round: keys 1 and 2 exchanged
    for k in ('xs',):
        d[k] = d[k][[0, 2, 1, 3]]
    d['bboxes'] = d['bboxes'][[0, 2, 1, 3]]
    res[1]['bbox'], res[2]['bbox'] = res[2]['bbox'], res[1]['bbox']
    res[1]['id'], res[2]['id'] = 2, 3
    acc = TR.Acc()
    TR.compare_stream('x', [(d, res)], ref, 0.4, [0.4], (acc,))
    r = acc.report()
    assert r['rank_swaps']['count'] == 1 and r['rank_swaps']['max_oracle_score_gap'] < 1e-5
    assert r['streams_with_id_divergence'] == 0 and r['streams_with_ids_permuted_by_a_birth_tie'] == 1


def test_threshold_flip_is_a_classified_divergence():
    ref = [_frame([0.9, 0.400002, 0.2], [1, 2, 0]), _frame([0.9, 0.5, 0.2], [1, 2, 0])]
    ours = [_frame([0.9, 0.399999, 0.2], [1, 0, 0]), _frame([0.9, 0.5, 0.2], [1, 2, 0])]
    acc = TR.Acc()
    TR.compare_stream('x', ours, ref, 0.4, [0.4], (acc,))
    r = acc.report()
    assert r['threshold_flips']['count'] == 1 and r['threshold_flips']['max_oracle_distance_from_threshold'] < 1e-5
    assert r['streams_with_id_divergence'] == 1 and r['frames_until_first_id_divergence'] == [0]
    assert not r['unexplained_divergences'] and r['frames_compared'] == 1


def test_unexplained_divergence_is_flagged():
    ref = [_frame([0.9, 0.7], [1, 2]), _frame([0.9, 0.7], [1, 2])]
    ours = [_frame([0.9, 0.7], [1, 2]), _frame([0.9, 0.7], [2, 1])]
    acc = TR.Acc()
    TR.compare_stream('x', ours, ref, 0.4, [0.4], (acc,))
    assert len(acc.report()['unexplained_divergences']) == 1


def test_prior_heatmap_flip_is_counted_and_its_next_frame_is_kept_apart():
    """a tracked object with score 0.5000001 vs 0.4999999 at pre_thresh = 0.5: rendered into the next prior heat-map on
    one side only; the deltas of the NEXT frame go into the "after_prior_flip" bucket"""
    ref = [_frame([0.9, 0.5000001, 0.2], [1, 2, 0]), _frame([0.9, 0.62, 0.2], [1, 2, 0]), _frame([0.9, 0.6, 0.2], [1, 2, 0])]
    ours = [_frame([0.9, 0.4999999, 0.2], [1, 2, 0]), _frame([0.9, 0.61, 0.2], [1, 2, 0]), _frame([0.9, 0.6, 0.2], [1, 2, 0])]
    acc = TR.Acc()
    TR.compare_stream('x', ours, ref, 0.4, [0.4, 0.5], (acc,), pre_thresh=0.5)
    r = acc.report()
    assert r['prior_heatmap_flips']['count'] == 1 and r['prior_heatmap_flips']['frames_compared_right_after_one'] == 1
    assert abs(r['prior_heatmap_flips']['abs_dscore_max_in_those_frames'] - 0.01) < 1e-6
    assert r['abs_dscore']['max'] < 1e-6 and r['streams_with_id_divergence'] == 0
