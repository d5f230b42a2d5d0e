"""CPU: association on REAL trajectories -- the KITTI tracking ground truth the reference ships for its evaluator
(src/tools/eval_kitti_track/data/tracking: ``label_02_val_half``, 21 videos, 4 030 frames, the split its KITTI experiment
is scored on; and ``label_02``, the 21 COMPLETE training videos, 8 029 frames) turned into detections and
tracked by the reference's own ``Tracker`` (tests/golden/make_kitti_gt.py -> tests/golden/kitti_gt_tracks.npz,
kitti_gt_tracks_full.npz).
The native C++ tracker (``ct_tracker_step_dets`` / ``ct_tracker_init_tracks``), the Python mirror
(``centertrack_amd/tracker.py``) and the oracle must return IDENTICAL (tracking_id, age, active, source detection)
lists, in the same order, on every frame of every mode: greedy / Hungarian, max_age 0 / 2, private / public
detections.  With /root/reference present the fixture's inputs are re-derived from the label files, every mode is
re-run through the reference class itself, and the reference's evaluator is run on files written by
``results_io.save_kitti_tracking_results`` (tools/eval_kitti_gt.py)."""
import json
import os
import sys
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_kitti_gt as G  # noqa: E402

from centertrack_amd import fast_track as FT  # noqa: E402
from centertrack_amd import tracker as TR  # noqa: E402
from oracle import tracker as OTR  # noqa: E402

REF = os.environ.get('CENTERTRACK_REFERENCE', '/root/reference')
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'src', 'tools', 'eval_kitti_track')),
                               reason='reference checkout absent (GPU box)')


FRAMES = {'val_half': 4030, 'full': 8029}          # the evaluator's own count (evaluate_tracking.py:107)


@pytest.fixture(scope='module', params=['val_half', 'full'])
def gold(request):
    g = dict(np.load(os.path.join(HERE, 'golden', G.SPLITS[request.param][2])))
    g['split'] = request.param
    return g


def _inputs(gold, kind):
    pub = gold.get(kind + '.public', np.zeros((0, 2), np.float32))
    pp = gold.get(kind + '.public_ptr', np.zeros(len(gold[kind + '.ptr']), np.int32))
    return gold[kind + '.dets'], gold[kind + '.ptr'], gold['seq_frames'], pub, pp


class _Native(object):
    """``Tracker`` face of the native tracker for G.run_tracker: dict items in, (id, age, active, src) out"""

    def __init__(self, new_thresh, max_age, hungarian, public_det):
        self.ft = FT.FastTracker(new_thresh, max_age, 100, hungarian=hungarian, public_det=public_det)

    def init_track(self, items):
        self.ft.init_tracks(items)

    def step(self, items, public_det=None):
        arr = FT.items_to_array(items)
        arr['row'] = np.arange(len(items))
        got = self.ft.step_dets(arr, public_det)
        return [{'tracking_id': int(r['tracking_id']), 'age': int(r['age']), 'active': int(r['active']),
                 'src': int(r['row'])} for r in got]


def _makers(kind, hung, public, max_age):
    nt = G.NEW_THRESH[kind]
    opt = types.SimpleNamespace(new_thresh=nt, max_age=max_age, hungarian=bool(hung), public_det=bool(public))
    return {'native': lambda: _Native(nt, max_age, bool(hung), bool(public)),
            'python': lambda: TR.Tracker(opt),
            'oracle': lambda: OTR.Tracker(nt, max_age, hungarian=bool(hung), public_det=bool(public))}


def _first_difference(got, gptr, want, wptr):
    for f in range(len(wptr) - 1):
        a, b = got[gptr[f]:gptr[f + 1]], want[wptr[f]:wptr[f + 1]]
        if a.shape != b.shape or (a != b).any():
            return 'frame %d: got %s want %s' % (f, a.tolist(), b.tolist())
    return 'pointer arrays differ'


@pytest.mark.parametrize('impl', ['native', 'python', 'oracle'])
@pytest.mark.parametrize('mode', [m[0] for m in G.MODES])
def test_ids_identical_to_the_reference_tracker_on_kitti_trajectories(gold, mode, impl):
    name, kind, hung, public, max_age = [m for m in G.MODES if m[0] == mode][0]
    dets, ptr, seq_frames, pub, pp = _inputs(gold, kind)
    assert int(seq_frames.sum()) == FRAMES[gold['split']] and len(ptr) == FRAMES[gold['split']] + 1
    rows, optr = G.run_tracker(_makers(kind, hung, public, max_age)[impl], dets, ptr, seq_frames, pub, pp, public)
    want, wptr = gold[name + '.tracks'], gold[name + '.ptr']
    same = rows.shape == want.shape and np.array_equal(optr, wptr) and np.array_equal(rows, want)
    assert same, _first_difference(rows, optr, want, wptr)
    # the streams exercise what they claim to
    if kind == 'noisy':
        assert (want[:, 2] > 1).sum() > 10000           # matches (active counts up)
        assert want[:, 0].max() > 50                     # births
    if max_age > 0:
        assert (want[:, 3] < 0).sum() > 1000             # tracks carried without a detection


def test_clean_streams_follow_the_ground_truth_identities(gold):
    """exact displacements, every box tracked.  Hungarian: a ground-truth trajectory keeps ONE tracking id for as
    long as it has consecutive boxes -- on val_half without exception; on the complete videos with ONE event (video 20,
    frame 413: the assignment has to place every old track, a newcomer with displacement 0 stands where a departed car's
    neighbour was, and the cheapest complete assignment is a chain of three thefts; the reference does exactly that).  Greedy (tracker.py:129-138 walks the detections in order and takes the
    nearest free track): the only identity changes are thefts -- an object WITHOUT a previous-frame box (displacement
    0) that comes earlier in the frame takes a neighbour's track, whose owner takes the next one or is re-born; the reference does
    exactly that, and every such event is checked to have that cause."""
    dets, ptr, seq_frames = gold['clean.dets'], gold['clean.ptr'], gold['seq_frames']
    counts = {}
    for mode in ('clean_greedy', 'clean_hungarian'):
        tracks, tptr = gold[mode + '.tracks'], gold[mode + '.ptr']
        f = 0
        switches = 0
        for n in seq_frames:
            last = {}
            for t in range(n):
                d, r = dets[ptr[f]:ptr[f + 1]], tracks[tptr[f]:tptr[f + 1]]
                assert len(r) == len(d)                  # every GT box is tracked (score 1 > new_thresh)
                cur = {int(d[src, 10]): int(tid) for tid, age, active, src in r}
                cls = {int(q[10]): int(q[1]) for q in d}
                moved = [g for g, tid in cur.items() if g in last and last[g] != tid]
                switches += len(moved)
                if moved:       # a chain of thefts starts at a newcomer of the same class holding an OLD id
                    starts = [g for g, tid in cur.items() if g not in last and tid in last.values()
                              and cls[g] in [cls[m] for m in moved]]
                    assert starts, (mode, f, moved)
                last = cur
                f += 1
        counts[mode] = switches
    want = {'val_half': (0, 20), 'full': (2, 80)}[gold['split']]
    assert counts['clean_hungarian'] == want[0] and 0 < counts['clean_greedy'] < want[1], counts


@needs_ref
def test_fixture_inputs_rederive_from_the_reference_labels(gold):
    seqs = G.read_sequences(REF, gold['split'])
    for kind in ('clean', 'noisy'):
        d, p, sf, pub, pp = G.synth_detections(seqs, kind)
        np.testing.assert_array_equal(d, gold[kind + '.dets'])
        np.testing.assert_array_equal(p, gold[kind + '.ptr'])
        if kind == 'noisy':
            np.testing.assert_array_equal(pub, gold['noisy.public'])


@needs_ref
@pytest.mark.parametrize('split', ['val_half', 'full'])
def test_reference_tracker_regenerates_the_fixture(split):
    """the reference's Tracker class itself (own process: ref_import stubs modules) on the re-derived inputs"""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(HERE, 'golden', 'make_kitti_gt.py'), '--split', split, '--check'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and 'CHECK OK' in p.stdout, p.stdout[-2000:]


@needs_ref
def test_reference_evaluator_on_files_written_by_results_io(tmp_path):
    """f2 hand-off: results_io.save_kitti_tracking_results -> the reference's evaluate_tracking.py (run unmodified
    as a subprocess in its own directory).  Clean GT detections: no miss, no false positive; Hungarian: no id switch ->
    MOTA 1.0 (the evaluator's ceiling) for both evaluated classes; the noisy stream's figures are pinned in tests/golden/kitti_gt_eval.json"""
    sys.path.insert(0, os.path.join(HERE, '..'))
    from tools import eval_kitti_gt as E
    res = E.evaluate(['clean_greedy', 'clean_hungarian', 'noisy_greedy_age2'], str(tmp_path), ref=REF)
    for cls in ('car', 'pedestrian'):
        s = res['clean_hungarian'][cls]
        assert s['MOTA'] == 1.0 and s['id-switches'] == 0 and s['fragmentations'] == 0, s
        for mode in ('clean_hungarian', 'clean_greedy'):        # greedy: the thefts of the test above, nothing else
            s = res[mode][cls]
            assert s['false positives'] == 0 and s['missed'] == 0 and s['recall'] == 1.0 and s['precision'] == 1.0, s
            assert s['id-switches'] <= 6 and s['MOTA'] > 0.998
    with open(os.path.join(HERE, 'golden', 'kitti_gt_eval.json')) as f:
        pinned = json.load(f)
    for mode in res:
        assert res[mode] == pinned[mode], (mode, res[mode], pinned[mode])


@needs_ref
def test_reference_evaluator_on_the_complete_videos(tmp_path):
    """the same hand-off on ``label_02`` (the evaluator run without a split argument, evaluate_tracking.py:975): nothing
    missed, nothing false; Hungarian: the one chain of thefts of the test above is the evaluator's only id switch"""
    sys.path.insert(0, os.path.join(HERE, '..'))
    from tools import eval_kitti_gt as E
    res = E.evaluate(['clean_hungarian', 'noisy_public_hungarian_age2'], str(tmp_path), ref=REF, split='full')
    for cls, ids in (('car', 1), ('pedestrian', 0)):
        s = res['clean_hungarian'][cls]
        assert s['false positives'] == 0 and s['missed'] == 0 and s['recall'] == 1.0 and s['precision'] == 1.0, s
        assert s['id-switches'] == ids and s['MOTA'] > 0.9999, s
    with open(os.path.join(HERE, 'golden', 'kitti_gt_eval_full.json')) as f:
        pinned = json.load(f)
    for mode in res:
        assert res[mode] == pinned[mode], (mode, res[mode], pinned[mode])
