"""GPU: the stated policy for the one way the 1e-3 value bar can be exceeded (README).

The reference renders the prior heat-map of frame t+1 from the tracks of frame t with the blob centre cast to an
integer (`ct_int = ct.astype(np.int32)`, src/lib/detector.py:281-283).  A track centre that lies within fp32
summation-order noise of an integer (470.00000 in the oracle, 469.99997 on the HIP path) moves its blob by one input
pixel.  Nothing about that frame is wrong on either side, but frame t+1 then runs on prior heat-maps that differ by one
pixel of one blob, and its scores differ by up to ~1e-2 before the streams re-converge (the next prior heat-map is
rendered from near-identical tracks again).  tools/tie_report.py measured it: once in 1664 frames
(profiles/r03_tie_report.json, `prior_heatmap_flips`), stream seed 39324 = `mot17_512 x1 run 39 stream 0`, frame 24.

Policy asserted here, on exactly that stream (T = 32, nothing re-seeded):
  * track ids: identical to the oracle's over the whole stream, flip or no flip;
  * a frame whose prior heat-map blobs equal the oracle's: every score within 1e-3, every box within the image-space
    tolerance of tests/_parity.py -- the north_star bar, unconditionally;
  * the frame right after a blob flip: exempt from the 1e-3 bar, bounded by 5e-2 (scores) / 0.5 px (boxes), same
    detections, same ids;
  * re-convergence: from the second frame after the flip on the bar is 1e-3 again (and the blobs agree again).
The flip itself depends on the last bits of the HIP path's sums (launch shapes).  It was found on round 3's plan; round
4's plan (Tree.project computed by the stride-2 conv launches: other summation orders in four layers) puts that centre on
the oracle's side of the integer and the same stream runs without a flip (profiles/r04_tie_report.json: 0 flips in 1760
frames).  The test therefore runs the stream on BOTH launch structures (``model.FUSE_PROJ = False`` = round 3's, its pinned
shapes are still in the table); through round 5 the flip reproduced on round 3's -- frame 24, blob (470, 344, 6) vs
(469, 344, 6) -- and exercised the exemption; since round 6 (other DCN arithmetic) neither structure flips on this stream.
Every frame has to meet the 1e-3 bar, or, wherever a flip occurs, the policy above."""
import os
import sys

import numpy as np
import pytest
import torch

from _parity import ATOL, calibrated_state_dict, scrolled_stream

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools'))

pytestmark = pytest.mark.gpu

SEED = 317 + 7 + 1000 * 39          # tools/tie_report.stream_seed(plan 0, run 39, stream 0)
T = 32


@pytest.mark.parametrize('plan', ['shipped', 'round3'])
def test_prior_heatmap_flip_stream_follows_the_stated_policy(device, plan, monkeypatch):
    import scenarios as S
    from centertrack_amd import model as M
    if plan == 'round3':
        monkeypatch.setattr(M, 'FUSE_PROJ', False)
    import tie_report as TR
    from centertrack_amd.detector import StreamDetector, default_opt
    from centertrack_amd.image import make_meta
    from centertrack_amd.model import DLASegHIP
    from oracle import detector as odet
    assert TR.stream_seed(0, 39, 0) == SEED
    name = 'mot17_512'
    cfg = S.CONFIGS[name]
    heads = S.HEAD_SETS[cfg['heads']]
    H, W = cfg['H'], cfg['W']
    sd = calibrated_state_dict(name, heads)
    kw = dict(track_thresh=cfg['track_thresh'], pre_thresh=cfg['pre_thresh'], flip_test=False)
    opt = default_opt(heads, **kw)
    model = DLASegHIP(heads)
    model.load_state_dict(sd)
    det = StreamDetector(opt, model=model, num_streams=1)
    oopt = odet.default_opt(input_h=H, input_w=W, num_classes=heads['hm'], **kw)
    oracle = odet.Detector(oopt, sd, heads)
    meta = make_meta(H, W, 2 * H, 2 * W)
    px_tol = ATOL * 2.0 * opt.down_ratio * 2 + 2e-3               # tests/_parity.py's image-space tolerance
    flips, exempt, worst, worst_exempt = [], [], 0.0, 0.0
    after_flip = False
    for t, img in enumerate(scrolled_stream(H, W, T, SEED)):
        got = TR._slim(det.results_as_dicts(det.step(img, [dict(meta)])[0], 0, meta))
        want = TR._slim(oracle.run(img, dict(meta)))
        post, after_flip = after_flip, False
        tol_s, tol_b = (5e-2, 0.5) if post else (ATOL, px_tol)
        assert len(got) == len(want), 'frame %d: %d results, oracle %d' % (t, len(got), len(want))
        gb = np.array([r['bbox'] for r in got], np.float64).reshape(-1, 4)
        used = set()
        for rw in want:
            wb = np.array(rw['bbox'], np.float64)
            cand = [i for i in range(len(got)) if i not in used and got[i]['class'] == rw['class']
                    and np.abs(gb[i] - wb).max() <= tol_b]
            assert len(cand) == 1, 'frame %d%s: oracle box %s has %d counterparts within %.3f px' % (
                t, ' (right after a blob flip)' if post else '', wb, len(cand), tol_b)
            rg = got[cand[0]]
            used.add(cand[0])
            ds = abs(rg['score'] - rw['score'])
            assert ds <= tol_s, 'frame %d%s: score %.6f vs oracle %.6f' % (t, ' (right after a blob flip)' if post else '',
                                                                            rg['score'], rw['score'])
            assert rg['id'] == rw['id'], 'frame %d: oracle track %d is our track %d' % (t, rw['id'], rg['id'])
            if post:
                worst_exempt = max(worst_exempt, ds)
            else:
                worst = max(worst, ds)
        if post:
            exempt.append(t)
        bo, bg = TR.prior_blobs(want, oopt.pre_thresh, meta), TR.prior_blobs(got, oopt.pre_thresh, meta)
        if bo != bg:
            # one blob, moved by one input pixel: the integer cast of a centre within fp32 noise of an integer
            only_o = [b for b in bo if b not in bg]
            only_g = [b for b in bg if b not in bo]
            assert len(only_o) == len(only_g) == 1, (t, only_o, only_g)
            assert max(abs(a - b) for a, b in zip(only_o[0], only_g[0])) == 1, (t, only_o, only_g)
            assert not post, 'frame %d: the blobs must agree again right after a flip' % t
            flips.append((t, only_o[0], only_g[0]))
            after_flip = True
    assert len(flips) <= 2, 'a blob flip is a rare event (0.6 per 1000 frames measured): %s' % (flips,)
    # Rounds 3-5: on round 3's launch structure this stream flipped at frame 24 -- blob (470, 344, 6) vs (469, 344, 6) -- and
    # frame 25 exercised the exemption (|dscore| 1.15e-2).  Round 6 changed the DCN arithmetic itself (Winograd offset convs,
    # v_exp / v_rcp mask sigmoid, re-measured schedules): the centre lands on the oracle's side of the integer on both launch
    # structures now and the stream runs without a flip (profiles/r06_gpu_tests.log).  Whatever flips, the policy above holds.
    if flips:
        assert exempt == [f[0] + 1 for f in flips if f[0] + 1 < T], (flips, exempt)
    print('tie policy [%s plan]: %d blob flip(s) %s; |dscore| max %.2e outside the exempt frames, %.2e inside (%s)'
          % (plan, len(flips), flips, worst, worst_exempt, exempt))
