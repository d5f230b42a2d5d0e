"""The id policy of tests/_parity.py on synthetic frames (CPU): what the stream-level GPU tests accept and what they refuse.

  identity                                   accepted
  two births of one rank-tie group swapped    accepted, enumerated (tracker.py:104-111 numbers new ids in rank order)
  two results of one rank-tie group exchange
  an existing id and a birth                  accepted, enumerated (round 6: greedy association serves detections in score order,
                                              tracker.py:56-76 -- tools/tie_report.py met it once in 1760 frames)
  the same exchange between scores 1e-3 apart refused
"""
import numpy as np
import pytest

from _parity import StreamParity


def _frame(scores, keys, ids, swap_rank=None, ids_ours=None):
    """decode dicts (ours, oracle) + result lists for detections at distinct cells; ``swap_rank`` = (i, j): our decode lists the
    two candidates in the other order"""
    n = len(scores)
    sc = np.array(scores, np.float32)
    od = {'scores': sc[None].copy(), 'clses': np.zeros((1, n), np.float32), 'ys': np.array([[k[0] for k in keys]], np.float32),
          'xs': np.array([[k[1] for k in keys]], np.float32)}
    order = list(range(n))
    if swap_rank:
        i, j = swap_rank
        order[i], order[j] = order[j], order[i]
    gd = {k: v[:, order].copy() for k, v in od.items()}

    def results(id_list):
        return [{'bbox': np.array([8.0 * k[1], 8.0 * k[0], 8.0 * k[1] + 40, 8.0 * k[0] + 40], np.float32), 'score': sc[i], 'class': 1,
                 'ct': np.array([8.0 * k[1] + 20, 8.0 * k[0] + 20], np.float32), 'tracking': np.zeros(2, np.float32), 'age': 1, 'active': 1,
                 'tracking_id': id_list[i]} for i, k in enumerate(keys)]
    return gd, od, results(ids_ours or ids), results(ids)


KEYS = [(10, 10), (30, 30), (50, 50), (70, 70), (90, 90), (20, 60)]
SCORES = [0.9, 0.8, 0.7, 0.600004, 0.6, 0.5]          # ranks 3 and 4 form a tie group (4e-6 apart)


def test_identity_passes():
    p = StreamParity('t')
    gd, od, got, want = _frame(SCORES, KEYS, [1, 2, 3, 4, 5, 6])
    assert p.check(0, gd, 0, od, got, want, 0.3, 8.0)
    assert p.finish() == []


def test_birth_tie_swap_is_enumerated():
    p = StreamParity('t')
    gd, od, got, want = _frame(SCORES, KEYS, [1, 2, 3, 4, 5, 6], swap_rank=(3, 4), ids_ours=[1, 2, 3, 5, 4, 6])
    p.check(0, gd, 0, od, got, want, 0.3, 8.0)
    assert p.finish() == [(4, 5), (5, 4)]


def test_exchange_of_an_existing_id_inside_a_rank_tie_group_is_enumerated():
    p = StreamParity('t')
    gd, od, got, want = _frame([0.9, 0.8, 0.7, 0.65, 0.5], KEYS[:5], [1, 2, 3, 4, 5])
    p.check(0, gd, 0, od, got, want, 0.3, 8.0)
    # next frame: the detections at ranks 3 / 4 tie; the oracle keeps id 4 on the first and gives birth to 7 on the second,
    # our side serves them in the other order
    gd, od, got, want = _frame(SCORES[:5], KEYS[:5], [1, 2, 3, 4, 7], swap_rank=(3, 4), ids_ours=[1, 2, 3, 7, 4])
    p.check(1, gd, 0, od, got, want, 0.3, 8.0)
    assert p.exchanges == [(1, 4, 7)]
    assert p.finish() == [(4, 7), (7, 4)]
    with pytest.raises(AssertionError):
        q = StreamParity('strict', strict=True)
        q.id_map, q.rev, q.tie_ids = dict(p.id_map), dict(p.rev), set(p.tie_ids)
        q.finish()


def test_exchange_between_scores_that_do_not_tie_is_refused():
    p = StreamParity('t')
    gd, od, got, want = _frame([0.9, 0.8, 0.7, 0.65, 0.5], KEYS[:5], [1, 2, 3, 4, 5])
    p.check(0, gd, 0, od, got, want, 0.3, 8.0)
    gd, od, got, want = _frame([0.9, 0.8, 0.7, 0.601, 0.6], KEYS[:5], [1, 2, 3, 4, 7], ids_ours=[1, 2, 3, 7, 4])
    with pytest.raises(AssertionError, match='no rank tie explains it'):
        p.check(1, gd, 0, od, got, want, 0.3, 8.0)
