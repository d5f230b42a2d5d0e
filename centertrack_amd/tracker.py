"""Displacement-based track association on the host (numpy), the step after decode.

Same interface and semantics as the reference ``Tracker`` (src/lib/utils/tracker.py:6-138):
``Tracker(opt)``, ``init_track(results)``, ``step(results, public_det=None)``, ``reset()``;
detections are dicts with ``score, class, ct, tracking, bbox`` and get ``tracking_id,
age, active``.  Track IDs must be bit-identical to the reference, so the float32 /
float64 promotion points are kept exactly: centres and sizes are float32, the squared
distances float32, and the ``+ invalid * 1e18`` gate promotes to float64
(tracker.py:44-50).  Greedy matching walks detections in their (score-descending) order and
takes the first minimum over tracks (tracker.py:129-138); ``--hungarian`` uses scipy's
linear_sum_assignment in place of the removed sklearn helper."""
import numpy as np

INVALID = 1e18


def greedy_assignment(dist):
    """rows = detections in score order, cols = tracks; consumes ``dist``."""
    pairs = []
    if dist.shape[1] > 0:
        for i in range(dist.shape[0]):
            j = int(dist[i].argmin())
            if dist[i, j] < 1e16:
                dist[:, j] = INVALID
                pairs.append((i, j))
    return np.array(pairs, np.int32).reshape(-1, 2)


def _area(boxes):
    b = np.array(boxes, np.float32).reshape(-1, 4) if len(boxes) else np.zeros((0, 4), np.float32)
    return ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(np.float32)


class Tracker(object):
    def __init__(self, opt):
        self.opt = opt
        self.reset()

    def reset(self):
        self.id_count = 0
        self.tracks = []

    def _birth(self, item):
        self.id_count += 1
        item['tracking_id'] = self.id_count
        item['age'] = 1
        item['active'] = 1

    def init_track(self, results):
        for item in results:
            if item['score'] > self.opt.new_thresh:
                self._birth(item)
                if 'ct' not in item:
                    bbox = item['bbox']
                    item['ct'] = [(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2]
                self.tracks.append(item)

    def step(self, results, public_det=None):
        N, M = len(results), len(self.tracks)
        # predicted position in the previous frame: ct + tracking (element-wise; ct may be a
        # python list in ddd mode, post_process.py:71-75, so go through numpy explicitly)
        dets = np.array([np.asarray(d['ct']) + np.asarray(d['tracking']) for d in results],
                        np.float32).reshape(N, 2)
        tracks = np.array([t['ct'] for t in self.tracks], np.float32).reshape(M, 2)
        track_size = _area([t['bbox'] for t in self.tracks])
        item_size = _area([d['bbox'] for d in results])
        track_cat = np.array([t['class'] for t in self.tracks], np.int32)
        item_cat = np.array([d['class'] for d in results], np.int32)
        dist = ((tracks.reshape(1, M, 2) - dets.reshape(N, 1, 2)) ** 2).sum(axis=2)
        invalid = ((dist > track_size.reshape(1, M)) + (dist > item_size.reshape(N, 1)) +
                   (item_cat.reshape(N, 1) != track_cat.reshape(1, M))) > 0
        dist = dist + invalid * INVALID
        if getattr(self.opt, 'hungarian', False):
            from scipy.optimize import linear_sum_assignment
            dist[dist > INVALID] = INVALID
            r, c = linear_sum_assignment(dist)
            cand = np.stack([r, c], axis=1)
            ok = dist[cand[:, 0], cand[:, 1]] <= 1e16 if len(cand) else np.zeros(0, bool)
            matches = cand[ok].reshape(-1, 2)
            all_matched_d = set(cand[:, 0].tolist())
            all_matched_t = set(cand[:, 1].tolist())
            unmatched_dets = [d for d in range(N) if d not in all_matched_d] + \
                [int(m[0]) for m, k in zip(cand, ok) if not k]
            unmatched_tracks = [t for t in range(M) if t not in all_matched_t] + \
                [int(m[1]) for m, k in zip(cand, ok) if not k]
        else:
            matches = greedy_assignment(dist.copy())
            md, mt = set(matches[:, 0].tolist()), set(matches[:, 1].tolist())
            unmatched_dets = [d for d in range(N) if d not in md]
            unmatched_tracks = [t for t in range(M) if t not in mt]
        ret = []
        for i, j in matches:
            track = results[i]
            track['tracking_id'] = self.tracks[j]['tracking_id']
            track['age'] = 1
            track['active'] = self.tracks[j]['active'] + 1
            ret.append(track)
        if getattr(self.opt, 'public_det', False) and len(unmatched_dets) > 0:
            # MOT public-detection protocol (tracker.py:83-101): births only next to a given det
            pub = np.array([d['ct'] for d in public_det], np.float32).reshape(-1, 2)
            dist3 = ((dets.reshape(-1, 1, 2) - pub.reshape(1, -1, 2)) ** 2).sum(axis=2)
            matched_dets = [d for d in range(N) if d not in unmatched_dets]
            dist3[matched_dets] = INVALID
            for j in range(len(pub)):
                i = int(dist3[:, j].argmin())
                if dist3[i, j] < item_size[i]:
                    dist3[i, :] = INVALID
                    if results[i]['score'] > self.opt.new_thresh:
                        self._birth(results[i])
                        ret.append(results[i])
        else:
            for i in unmatched_dets:
                if results[i]['score'] > self.opt.new_thresh:
                    self._birth(results[i])
                    ret.append(results[i])
        for j in unmatched_tracks:
            track = self.tracks[j]
            if track['age'] < self.opt.max_age:
                track['age'] += 1
                track['active'] = 0
                bbox, ct = track['bbox'], track['ct']
                track['bbox'] = [bbox[0] + 0, bbox[1] + 0, bbox[2] + 0, bbox[3] + 0]
                track['ct'] = [ct[0] + 0, ct[1] + 0]
                ret.append(track)
        self.tracks = ret
        return ret
