"""Shared checker of the stream-level parity tests (HIP path vs the CPU oracle, frame by frame).

north_star's bar, as asserted here:
  * top-K indices (class, y, x), classes and ranks: IDENTICAL for every detection whose oracle score is >= out_thresh,
    except inside a *tie group* -- consecutive oracle ranks whose scores differ by less than ``TIE`` (1e-5; ``RANK_TIE_UNPICKED``
    = 5e-5 on the streams nobody picked, tests/test_hip_plans.py: the measured width of fp32 rank noise): fp32 with
    a different summation order cannot resolve those, they may swap among themselves;
  * heat-map scores and every decode-level value (boxes, centres, tracking displacement, 3D heads) on the OUTPUT
    GRID: within ``ATOL`` = 1e-3 absolute (depth, an unbounded 1/sigmoid - 1: 1e-3 relative on top);
  * image-space results (after the inverse affine): within 2 x ATOL x (image px per output cell) + 2e-3 px of fp32 slack
    -- a box edge is centre + reg -+ wh / 2 (or + an ltrb_amodal entry), i.e. up to 1.5 ATOL on the grid when every head
    value is within ATOL;
  * a detection whose oracle score lies within ``TIE`` of a threshold (out / new-track / prior-heat-map) is a
    *threshold tie*: which side it falls on is as unresolvable as a rank tie, and since it adds or removes a result --
    possibly a birth, shifting every later id -- the hand-picked streams of tests/test_hip_fullsize.py contain none and
    ``check`` refuses one that does (a statement about the test data, made from the ORACLE's scores alone); the streams
    of tests/test_hip_plans.py (default seeds, nobody picked them) are compared up to the frame before such a tie
    (``on_threshold_tie='stop'``), and tools/tie_report.py MEASURES ties instead of avoiding them: how often they occur
    per 1000 frames and what the HIP path does when they do (profiles/r03_tie_report.json);
  * track IDs: a consistent bijection oracle-id <-> our-id over the WHOLE stream that is the identity, except for ids
    handed out inside one birth tie group (new ids are numbered in rank order, tracker.py:104-111, so a tie swap of
    two births swaps their ids for the rest of the stream) and for two results of one rank-tie group that exchange
    their ids (greedy association serves detections in score order, tracker.py:56-76: which of two tied detections
    takes a track both can reach follows their rank swap).  Every non-identity pair is enumerated and must be
    explained by such a tie; ``strict`` demands the identity.
"""
import numpy as np

TIE = 1e-5
# rank-tie width of the streams NOBODY PICKED (tests/test_hip_plans.py): tools/tie_report.py measured, over 1664 full-size
# frames with nothing re-seeded, 14 rank swaps between detections whose oracle scores lie up to 4e-5 apart
# (profiles/r03_tie_report.json, `rank_swaps.max_oracle_score_gap`) -- fp32 noise of a 50-layer network on a sigmoid
# score is ~1e-5 (|dscore| max 1.4e-5 in tests/test_hip_tie_policy.py), so two scores 2-4e-5 apart can land either way
# round.  The hand-picked streams of the full-size tests keep the 1e-5 width.
RANK_TIE_UNPICKED = 5e-5
ATOL = 1e-3
GRID_FIELDS = ('bboxes', 'bboxes_amodal', 'tracking', 'rot', 'dim', 'amodel_offset', 'nuscenes_att', 'velocity')


class StreamParity(object):
    def __init__(self, tag, strict=False, on_threshold_tie='refuse', rank_tie=TIE):
        self.tag = tag
        self.strict = strict
        self.rank_tie = rank_tie    # consecutive oracle ranks closer than this form one tie group (may swap among themselves)
        # 'refuse': a threshold tie is a defect of the TEST DATA (hand-picked streams); 'stop': the stream is compared
        # up to the frame before the tie and the event is recorded in ``stopped`` (streams nobody picked: the plans
        # that are benchmarked, tools/tie_report.py measures how often this happens and what the HIP path does then)
        self.on_threshold_tie = on_threshold_tie
        self.stopped = None         # (frame, threshold, distance) of the threshold tie that ended the comparison
        self.id_map = {}            # oracle id -> our id
        self.rev = {}
        self.tie_ids = set()        # oracle ids born inside a tie group (the only ones allowed to map off-identity)
        self.frames = 0
        self.detections = 0
        # order changes inside tie groups that only exist because rank_tie is wider than TIE (consecutive oracle scores
        # 1e-5 .. rank_tie apart): tolerated one by one, but COUNTED -- a decode that mis-orders close scores
        # systematically would swap in every frame, fp32 noise does in < 1 % of them (advisor, round 4)
        self.wide_swaps = []
        self.exchanges = []         # (frame, oracle id, oracle id): ids exchanged between two results of one rank-tie group

    @staticmethod
    def _key(d, b, i):
        return (int(d['clses'][b, i]), int(d['ys'][b, i]), int(d['xs'][b, i]))

    def check(self, t, gd, gb, od, got, want, out_thresh, px_per_cell, min_dets=1, thresholds=()):
        """gd / od: our / the oracle's decode dict ([B,K,...] numpy; ours at batch index ``gb``, the oracle's at 0);
        got / want: result lists (dicts) of the frame; px_per_cell: image pixels per output-grid cell; thresholds: the
        other score thresholds of the run (new_thresh, pre_thresh)."""
        tag = '%s frame %d' % (self.tag, t)
        if self.stopped is not None:
            return False
        sc = od['scores'][0]
        for th in sorted(set((out_thresh,) + tuple(thresholds))):
            edge = float(np.abs(sc.astype(np.float64) - th).min())
            if edge < TIE and self.on_threshold_tie == 'stop':
                self.stopped = (t, float(th), edge)
                return False
            assert edge >= TIE, ('%s: TEST DATA: an oracle score lies %.1e from the threshold %.3f (a threshold tie, see the '
                                 'module docstring): pick another stream seed' % (tag, edge, th))
        n = int((sc >= out_thresh).sum())
        assert n >= min_dets, '%s: the synthetic stream must produce detections (%d)' % (tag, n)
        np.testing.assert_allclose(gd['scores'][gb, :n], sc[:n], atol=ATOL, err_msg=tag + ' scores')
        groups, a = [], 0
        for i in range(1, n + 1):
            if i == n or sc[i - 1] - sc[i] >= self.rank_tie:
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
                if self.rank_tie > TIE and [self._key(gd, gb, i) for i in range(a, b)] != [self._key(od, 0, i) for i in range(a, b)] \
                        and max(float(sc[i - 1] - sc[i]) for i in range(a + 1, b)) >= TIE:
                    self.wide_swaps.append((t, a, b - 1, float(sc[a] - sc[b - 1])))
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
        tol = ATOL * px_per_cell * 2 + 2e-3      # 1e-3 of a cell on either edge of a box / point pair + fp32 slack
        gbox = np.array([np.asarray(r['bbox'], np.float64) for r in got]).reshape(-1, 4)
        used = set()
        pairs = []
        for rw in want:
            wb = np.asarray(rw['bbox'], np.float64)
            cand = [i for i in range(len(got)) if i not in used and int(got[i]['class']) == int(rw['class'])
                    and np.abs(gbox[i] - wb).max() <= tol]
            assert len(cand) == 1, '%s: oracle result %s has %d counterparts within %.1e px' % (tag, wb, len(cand), tol)
            rg = got[cand[0]]
            used.add(cand[0])
            pairs.append((rw, rg))
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

        def tied_group(r):
            """index of the rank-tie group (> 1 member) the detection behind oracle result ``r`` sits in, else None"""
            hit = np.nonzero(sc[:n] == np.float32(np.asarray(r['score'])))[0]
            if not len(hit) or self._key(od, 0, int(hit[0])) not in tied_keys:
                return None
            return next(gi for gi, (a, b) in enumerate(groups) if a <= int(hit[0]) < b)

        for rw, rg in pairs:
            wid, gid = int(rw['tracking_id']), int(rg['tracking_id'])
            if wid in self.id_map and self.id_map[wid] == gid:
                continue
            if wid not in self.id_map and gid not in self.rev:      # a birth (or the first sighting of an id)
                self.id_map[wid], self.rev[gid] = gid, wid
                # born this frame from a detection inside a tie group?
                if tied_group(rw) is not None:
                    self.tie_ids.add(wid)
                continue
            # Two results of ONE rank-tie group exchanged their ids: greedy association (tracker.py:56-76) serves the detections in
            # score order, so which of two tied detections takes a track both can reach -- and which one is born, or takes the
            # second-best track -- follows their rank swap (round 6: once in 1760 frames, profiles/r06_tie_report.json).  The other
            # result of the exchange is the one that carries the id this one was expected to carry.
            other = [(w2, g2) for w2, g2 in pairs if w2 is not rw and
                     (int(g2['tracking_id']) == self.id_map[wid] if wid in self.id_map else int(w2['tracking_id']) == self.rev[gid])]
            gi = tied_group(rw)
            assert len(other) == 1 and gi is not None and tied_group(other[0][0]) == gi, (
                '%s: oracle track %d is our track %d, was %s -- and no rank tie explains it' % (tag, wid, gid, self.id_map.get(wid)))
            w2, g2 = int(other[0][0]['tracking_id']), int(other[0][1]['tracking_id'])
            for w_, g_ in ((wid, gid), (w2, g2)):
                self.id_map[w_], self.rev[g_] = g_, w_
            self.tie_ids.update((wid, w2))
            self.exchanges.append((t, wid, w2))
        self.frames += 1
        self.detections += len(want)
        return True

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


def setup_config(name, streams, **optkw):
    """model + StreamDetector + one CPU oracle per stream for the BASELINE configuration ``name`` (scenarios.CONFIGS) with
    ``streams`` streams per GPU: seeded random weights with the per-class heat-map calibration (scores spread, classes
    mixed, ~20-90 detections above the threshold) and 6-cell boxes so that consecutive frames associate"""
    import scenarios as S
    from centertrack_amd.detector import StreamDetector, default_opt
    from centertrack_amd.image import make_meta
    from centertrack_amd.model import DLASegHIP
    from oracle import detector as odet
    cfg = S.CONFIGS[name]
    heads = S.HEAD_SETS[cfg['heads']]
    H, W = cfg['H'], cfg['W']
    sd = calibrated_state_dict(name, heads)
    kw = dict(track_thresh=cfg['track_thresh'], pre_thresh=cfg['pre_thresh'], flip_test=cfg['flip'])
    kw.update(optkw)
    opt = default_opt(heads, **kw)
    model = DLASegHIP(heads)
    model.load_state_dict(sd)
    det = StreamDetector(opt, model=model, num_streams=streams)
    oopt = odet.default_opt(input_h=H, input_w=W, num_classes=heads['hm'], **kw)
    oracles = [odet.Detector(oopt, sd, heads) for _ in range(streams)]
    meta = make_meta(H, W, 2 * H, 2 * W)
    px_per_cell = 2.0 * opt.down_ratio                       # image = 2x the network input; grid = input / 4
    return cfg, opt, oopt, model, det, oracles, meta, px_per_cell


def run_config(name, streams, T, strict=False, min_tracks=5, seed0=317 + 7, sample=None, on_threshold_tie='refuse',
               rank_tie=TIE, **kw):
    """``streams`` streams advance T frames through ONE StreamDetector (the launch plan of that stream count); the
    streams in ``sample`` (default: all) are compared with the CPU oracle frame by frame.  Returns (checks, swaps, det)."""
    import torch
    cfg, opt, oopt, model, det, oracles, meta, ppc = setup_config(name, streams, **kw)
    H, W = cfg['H'], cfg['W']
    sample = list(range(streams)) if sample is None else list(sample)
    frames = [scrolled_stream(H, W, T, seed0 + 100 * s) for s in range(streams)]
    checks = {s: StreamParity('%s x%d stream %d' % (name, streams, s), strict=strict, on_threshold_tie=on_threshold_tie,
                              rank_tie=rank_tie)
              for s in sample}
    for t in range(T):
        res = det.step(torch.cat([frames[s][t] for s in range(streams)], 0), [dict(meta) for _ in range(streams)])
        gd = det.last_dets
        for s in sample:
            if checks[s].stopped is not None:
                continue
            img = frames[s][t]
            want = oracles[s].run(torch.cat((img, torch.flip(img, [3])), 0) if cfg['flip'] else img, dict(meta))
            got = det.results_as_dicts(res[s], s, meta)
            checks[s].check(t, gd, s, oracles[s].last_dets, got, want, oopt.out_thresh, ppc, min_dets=5,
                            thresholds=(oopt.new_thresh, oopt.pre_thresh))
    swaps = [checks[s].finish(min_tracks=min_tracks if checks[s].stopped is None else 0) for s in sample]
    return [checks[s] for s in sample], swaps, det
