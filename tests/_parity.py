"""Shared checker of the stream-level parity tests (HIP path vs the CPU oracle, frame by frame).

north_star's bar, as asserted here:
  * top-K indices (class, y, x), classes and ranks: IDENTICAL for every detection whose oracle score is >= out_thresh,
    except inside a *tie group* -- consecutive oracle ranks whose scores differ by less than ``TIE`` (1e-5): fp32 with
    a different summation order cannot resolve those, they may swap among themselves;
  * heat-map scores and every decode-level value (boxes, centres, tracking displacement, 3D heads) on the OUTPUT
    GRID: within ``ATOL`` = 1e-3 absolute (depth, an unbounded 1/sigmoid - 1: 1e-3 relative on top);
  * image-space results (after the inverse affine): within ATOL x (image px per output cell) + fp32 slack;
  * a detection whose oracle score lies within ``TIE`` of a threshold (out / new-track / prior-heat-map) is a
    *threshold tie*: which side it falls on is as unresolvable as a rank tie, and since it adds or removes a result --
    possibly a birth, shifting every later id -- the synthetic streams are chosen to contain none; ``check`` refuses a
    stream that has one (that is a statement about the test data, made from the ORACLE's scores alone);
  * track IDs: a consistent bijection oracle-id <-> our-id over the WHOLE stream that is the identity, except for ids
    handed out inside one birth tie group (new ids are numbered in rank order, tracker.py:104-111, so a tie swap of
    two births swaps their ids for the rest of the stream).  Every non-identity pair is enumerated and must be
    explained by such a tie; ``strict`` demands the identity.
"""
import numpy as np

TIE = 1e-5
ATOL = 1e-3
GRID_FIELDS = ('bboxes', 'bboxes_amodal', 'tracking', 'rot', 'dim', 'amodel_offset', 'nuscenes_att', 'velocity')


class StreamParity(object):
    def __init__(self, tag, strict=False):
        self.tag = tag
        self.strict = strict
        self.id_map = {}            # oracle id -> our id
        self.rev = {}
        self.tie_ids = set()        # oracle ids born inside a tie group (the only ones allowed to map off-identity)
        self.frames = 0
        self.detections = 0

    @staticmethod
    def _key(d, b, i):
        return (int(d['clses'][b, i]), int(d['ys'][b, i]), int(d['xs'][b, i]))

    def check(self, t, gd, gb, od, got, want, out_thresh, px_per_cell, min_dets=1, thresholds=()):
        """gd / od: our / the oracle's decode dict ([B,K,...] numpy; ours at batch index ``gb``, the oracle's at 0);
        got / want: result lists (dicts) of the frame; px_per_cell: image pixels per output-grid cell; thresholds: the
        other score thresholds of the run (new_thresh, pre_thresh)."""
        tag = '%s frame %d' % (self.tag, t)
        sc = od['scores'][0]
        for th in set((out_thresh,) + tuple(thresholds)):
            edge = float(np.abs(sc.astype(np.float64) - th).min())
            assert edge >= TIE, ('%s: TEST DATA: an oracle score lies %.1e from the threshold %.3f (a threshold tie, see the '
                                 'module docstring): pick another stream seed' % (tag, edge, th))
        n = int((sc >= out_thresh).sum())
        assert n >= min_dets, '%s: the synthetic stream must produce detections (%d)' % (tag, n)
        np.testing.assert_allclose(gd['scores'][gb, :n], sc[:n], atol=ATOL, err_msg=tag + ' scores')
        groups, a = [], 0
        for i in range(1, n + 1):
            if i == n or sc[i - 1] - sc[i] >= TIE:
                groups.append((a, i))
                a = i
        assert len(groups) >= 0.75 * n, '%s: degenerate stream (mostly near-ties): %d groups of %d' % (tag, len(groups), n)
        tied_keys = set()
        for a, b in groups:
            ours = sorted(self._key(gd, gb, i) for i in range(a, b))
            ref = sorted(self._key(od, 0, i) for i in range(a, b))
            assert ours == ref, '%s: top-K entries at ranks %d..%d differ\n got %s\nwant %s' % (tag, a, b - 1, ours, ref)
            if b - a > 1:
                tied_keys.update(ref)
        # (also below the threshold cut the rank of the first n entries is what the tracker sees: nothing else matters)
        by_key = {self._key(gd, gb, i): i for i in range(n)}
        for j in range(n):                                         # decode-level values on the output grid
            i = by_key[self._key(od, 0, j)]
            for f in GRID_FIELDS:
                if f in od:
                    np.testing.assert_allclose(gd[f][gb, i], od[f][0, j], rtol=0, atol=ATOL,
                                               err_msg='%s %s of rank %d' % (tag, f, j))
            if 'dep' in od:
                np.testing.assert_allclose(gd['dep'][gb, i], od['dep'][0, j], rtol=1e-3, atol=ATOL,
                                           err_msg='%s dep of rank %d' % (tag, j))
        # ---- image-space results + track ids ----
        assert len(got) == len(want), '%s: %d results, oracle %d' % (tag, len(got), len(want))
        tol = ATOL * px_per_cell * 2 + 2e-3
        gbox = np.array([np.asarray(r['bbox'], np.float64) for r in got]).reshape(-1, 4)
        used = set()
        for rw in want:
            wb = np.asarray(rw['bbox'], np.float64)
            cand = [i for i in range(len(got)) if i not in used and int(got[i]['class']) == int(rw['class'])
                    and np.abs(gbox[i] - wb).max() <= tol]
            assert len(cand) == 1, '%s: oracle result %s has %d counterparts within %.1e px' % (tag, wb, len(cand), tol)
            rg = got[cand[0]]
            used.add(cand[0])
            np.testing.assert_allclose(float(rg['score']), float(np.asarray(rw['score'])), atol=ATOL)
            for k in ('ct', 'tracking'):
                np.testing.assert_allclose(np.asarray(rg[k], np.float64), np.asarray(rw[k], np.float64), atol=tol,
                                           err_msg='%s %s' % (tag, k))
            assert int(rg['age']) == int(rw['age']) and int(rg['active']) == int(rw['active']), tag
            for k in ('dep', 'dim', 'alpha', 'loc', 'rot_y'):
                if k in rw:
                    np.testing.assert_allclose(np.asarray(rg[k], np.float64).reshape(-1),
                                               np.asarray(rw[k], np.float64).reshape(-1), rtol=2e-3, atol=2e-3,
                                               err_msg='%s %s' % (tag, k))
            wid, gid = int(rw['tracking_id']), int(rg['tracking_id'])
            if wid not in self.id_map:                             # a birth (or the first sighting of an id)
                assert gid not in self.rev, '%s: our id %d already stands for oracle id %d' % (tag, gid, self.rev.get(gid))
                self.id_map[wid], self.rev[gid] = gid, wid
                hit = np.nonzero(sc[:n] == np.float32(np.asarray(rw['score'])))[0]
                # born this frame from the detection at oracle rank hit[0]: inside a tie group?
                if len(hit) and self._key(od, 0, int(hit[0])) in tied_keys:
                    self.tie_ids.add(wid)
            assert self.id_map[wid] == gid, '%s: oracle track %d is our track %d, was %d' % (tag, wid, gid, self.id_map[wid])
        self.frames += 1
        self.detections += len(want)

    def finish(self, min_tracks=1):
        """every off-identity id pair must come from a birth tie group; returns the enumerated swaps"""
        swaps = sorted((w, g) for w, g in self.id_map.items() if w != g)
        assert len(self.id_map) >= min_tracks, '%s: only %d tracks were exercised' % (self.tag, len(self.id_map))
        unexplained = [(w, g) for w, g in swaps if w not in self.tie_ids]
        assert not unexplained, '%s: track ids differ without a score tie behind them: %s' % (self.tag, unexplained)
        if self.strict:
            assert not swaps, '%s: ids must be identical, swaps %s' % (self.tag, swaps)
        return swaps


def scrolled_stream(H, W, T, seed, step=4):
    """T frames [1,3,H,W]: a fixed N(0,1) image scrolled by ``step`` input px per frame -- every response of the
    network moves one output cell per frame: blobs enter at one edge, drift across and leave at the other, so
    births, associations and deaths all occur"""
    import torch
    g = torch.Generator().manual_seed(seed)
    base = torch.randn((3, H, W + step * T), generator=g, dtype=torch.float64).float()
    return [base[:, :, step * t:step * t + W].contiguous().unsqueeze(0) for t in range(T)]


def calibrated_state_dict(name, heads, box_cells=6.0):
    """The synthetic weights of the full-size parity streams: seed 317, with the per-class heat-map calibration of
    tests/golden/hm_calibration.json (scores spread below 0.9, ~40 detections above the threshold, classes mixed;
    made by tests/golden/make_hm_calibration.py with the CPU oracle) and boxes of ``box_cells`` output cells so that
    the tracker's size gate (dist^2 < box area, tracker.py:47-48) lets consecutive frames associate."""
    import json
    import os

    import torch
    from centertrack_amd import weights as Wt
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'hm_calibration.json')) as f:
        cal = json.load(f)[name]
    sd = Wt.make_synthetic_state_dict(heads, seed=317, hm_gain=1.0)
    s = torch.tensor(cal['scale'], dtype=torch.float64)
    sd['hm.2.weight'] = (sd['hm.2.weight'].double() * s.view(-1, 1, 1, 1)).float()
    sd['hm.2.bias'] = torch.tensor(cal['bias'], dtype=torch.float64).float()
    half = box_cells / 2
    if 'ltrb_amodal' in heads:
        sd['ltrb_amodal.2.bias'] = torch.tensor([-half, -half, half, half])
    sd['wh.2.bias'] = torch.tensor([box_cells, box_cells])
    return sd
