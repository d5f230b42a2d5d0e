"""Oracle: displacement-based track association on the CPU (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the behaviour of the reference's
``src/lib/utils/tracker.py`` -- ``Tracker.init_track`` (:11-22), ``Tracker.step`` (:28-127) and
``greedy_assignment`` (:129-138) -- as small functions over numpy arrays; pinned against the imported
reference by tests/golden/make_golden.py (tracker.json).  ``sklearn.utils.linear_assignment_`` (gone from
modern sklearn) is replaced by scipy's ``linear_sum_assignment`` for the ``--hungarian`` branch.

Semantics that matter for bit-identical IDs:
  * cost[i, j] = float32 squared distance between (ct + tracking) of detection i and ct of track j; a pair
    is gated out (cost + 1e18, which promotes the matrix to float64) when the distance exceeds either box
    area or the classes differ;
  * greedy: detections in input (= score) order, each takes the first minimum of its row if < 1e16 and
    retires that column;
  * output order: matched detections (detection order), then new tracks (detection order, ids counted
    up), then surviving unmatched tracks (track order).
"""
import numpy as np


def _box_area(items):
    return np.array([(b['bbox'][2] - b['bbox'][0]) * (b['bbox'][3] - b['bbox'][1]) for b in items], np.float32)


def _classes(items):
    return np.array([b['class'] for b in items], np.int32)


def cost_matrix(dets, tracks):
    """float64 [N, M] gated cost and the float32 predicted centres / detection areas it was built from."""
    n, m = len(dets), len(tracks)
    moved = np.array([d['ct'] + d['tracking'] for d in dets], np.float32)
    prev = np.array([t['ct'] for t in tracks], np.float32)
    d2 = ((prev.reshape(1, -1, 2) - moved.reshape(-1, 1, 2)) ** 2).sum(axis=2)
    det_area, trk_area = _box_area(dets), _box_area(tracks)
    gated = ((d2 > trk_area.reshape(1, m)) + (d2 > det_area.reshape(n, 1)) +
             (_classes(dets).reshape(n, 1) != _classes(tracks).reshape(1, m))) > 0
    return d2 + gated * 1e18, moved, det_area


def greedy_assignment(dist):
    """pairs [[det, track], ...]; mutates ``dist`` (matched columns are retired with 1e18)"""
    pairs = []
    if dist.shape[1] > 0:
        for det in range(dist.shape[0]):
            trk = dist[det].argmin()
            if dist[det][trk] < 1e16:
                pairs.append([det, trk])
                dist[:, trk] = 1e18
    return np.array(pairs, np.int32).reshape(-1, 2)


def hungarian_assignment(dist):
    from scipy.optimize import linear_sum_assignment
    dist[dist > 1e18] = 1e18
    rows, cols = linear_sum_assignment(dist)
    return np.stack([rows, cols], axis=1)


class Tracker(object):
    def __init__(self, new_thresh, max_age=-1, hungarian=False, public_det=False):
        self.new_thresh, self.max_age = new_thresh, max_age
        self.hungarian, self.public_det = hungarian, public_det
        self.reset()

    def reset(self):
        self.id_count, self.tracks = 0, []

    def _birth(self, det):
        """a detection becomes a new track if it is confident enough; returns it or None"""
        if not det['score'] > self.new_thresh:
            return None
        self.id_count += 1
        det.update(tracking_id=self.id_count, age=1, active=1)
        return det

    def init_track(self, results):
        for det in results:
            if self._birth(det) is not None:
                if 'ct' not in det:
                    x0, y0, x1, y1 = det['bbox']
                    det['ct'] = [(x0 + x1) / 2, (y0 + y1) / 2]
                self.tracks.append(det)

    def step(self, results, public_det=None):
        dist, moved, det_area = cost_matrix(results, self.tracks)
        n, m = len(results), len(self.tracks)
        pairs = hungarian_assignment(dist) if self.hungarian else greedy_assignment(dist.copy())
        lone_dets = [i for i in range(n) if i not in pairs[:, 0]]
        lone_tracks = [j for j in range(m) if j not in pairs[:, 1]]
        if self.hungarian:                       # optimal pairs may still be gated out
            kept = []
            for i, j in pairs:
                if dist[i, j] > 1e16:
                    lone_dets.append(i)
                    lone_tracks.append(j)
                else:
                    kept.append([i, j])
            pairs = np.array(kept).reshape(-1, 2)
        out = []
        for i, j in pairs:                       # continued tracks keep their id
            det, old = results[i], self.tracks[j]
            det.update(tracking_id=old['tracking_id'], age=1, active=old['active'] + 1)
            out.append(det)
        if self.public_det and len(lone_dets) > 0:
            out.extend(self._births_near_public(results, moved, det_area, lone_dets, public_det))
        else:
            out.extend(t for t in (self._birth(results[i]) for i in lone_dets) if t is not None)
        for j in lone_tracks:                    # unmatched tracks coast while young enough
            old = self.tracks[j]
            if old['age'] < self.max_age:
                old['age'] += 1
                old['active'] = 0
                old['bbox'] = [old['bbox'][0] + 0, old['bbox'][1] + 0, old['bbox'][2] + 0, old['bbox'][3] + 0]
                old['ct'] = [old['ct'][0] + 0, old['ct'][1] + 0]
                out.append(old)
        self.tracks = out
        return out

    def _births_near_public(self, results, moved, det_area, lone_dets, public_det):
        """MOT public-detection protocol: a track may only start at the unmatched detection nearest to a public
        detection, and only if that distance is below the detection's area."""
        pub = np.array([p['ct'] for p in public_det], np.float32)
        d3 = ((moved.reshape(-1, 1, 2) - pub.reshape(1, -1, 2)) ** 2).sum(axis=2)
        d3[[i for i in range(moved.shape[0]) if i not in lone_dets]] = 1e18
        born = []
        for col in range(len(pub)):
            i = d3[:, col].argmin()
            if d3[i, col] < det_area[i]:
                d3[i, :] = 1e18
                t = self._birth(results[i])
                if t is not None:
                    born.append(t)
        return born
