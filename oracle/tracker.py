"""Oracle: displacement-based greedy/Hungarian track association (numpy).
TEST INFRASTRUCTURE ONLY.  Restates ``src/lib/utils/tracker.py`` (Tracker :6-127,
greedy_assignment :129-138).  ``sklearn.utils.linear_assignment_`` (removed from
modern sklearn) is restated with scipy's linear_sum_assignment, as the reference's
Hungarian branch only needs the optimal pairs.
"""
import copy

import numpy as np


def linear_assignment(cost):
    from scipy.optimize import linear_sum_assignment
    r, c = linear_sum_assignment(cost)
    return np.stack([r, c], axis=1)


def greedy_assignment(dist):
    """tracker.py:129-138"""
    matched = []
    if dist.shape[1] == 0:
        return np.array(matched, np.int32).reshape(-1, 2)
    for i in range(dist.shape[0]):
        j = dist[i].argmin()
        if dist[i][j] < 1e16:
            dist[:, j] = 1e18
            matched.append([i, j])
    return np.array(matched, np.int32).reshape(-1, 2)


class Tracker(object):
    def __init__(self, new_thresh, max_age=-1, hungarian=False, public_det=False):
        self.new_thresh = new_thresh
        self.max_age = max_age
        self.hungarian = hungarian
        self.public_det = public_det
        self.reset()

    def init_track(self, results):
        """tracker.py:11-22"""
        for item in results:
            if item['score'] > self.new_thresh:
                self.id_count += 1
                item['active'] = 1
                item['age'] = 1
                item['tracking_id'] = self.id_count
                if 'ct' not in item:
                    bbox = item['bbox']
                    item['ct'] = [(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2]
                self.tracks.append(item)

    def reset(self):
        self.id_count = 0
        self.tracks = []

    def step(self, results, public_det=None):
        """tracker.py:28-127"""
        N = len(results)
        M = len(self.tracks)
        dets = np.array([det['ct'] + det['tracking'] for det in results], np.float32)
        track_size = np.array([((t['bbox'][2] - t['bbox'][0]) * (t['bbox'][3] - t['bbox'][1]))
                               for t in self.tracks], np.float32)
        track_cat = np.array([t['class'] for t in self.tracks], np.int32)
        item_size = np.array([((it['bbox'][2] - it['bbox'][0]) * (it['bbox'][3] - it['bbox'][1]))
                              for it in results], np.float32)
        item_cat = np.array([it['class'] for it in results], np.int32)
        tracks = np.array([pre['ct'] for pre in self.tracks], np.float32)
        dist = (((tracks.reshape(1, -1, 2) - dets.reshape(-1, 1, 2)) ** 2).sum(axis=2))
        invalid = ((dist > track_size.reshape(1, M)) + (dist > item_size.reshape(N, 1)) +
                   (item_cat.reshape(N, 1) != track_cat.reshape(1, M))) > 0
        dist = dist + invalid * 1e18
        if self.hungarian:
            dist[dist > 1e18] = 1e18
            matched_indices = linear_assignment(dist)
        else:
            matched_indices = greedy_assignment(copy.deepcopy(dist))
        unmatched_dets = [d for d in range(dets.shape[0]) if not (d in matched_indices[:, 0])]
        unmatched_tracks = [d for d in range(tracks.shape[0]) if not (d in matched_indices[:, 1])]
        if self.hungarian:
            matches = []
            for m in matched_indices:
                if dist[m[0], m[1]] > 1e16:
                    unmatched_dets.append(m[0])
                    unmatched_tracks.append(m[1])
                else:
                    matches.append(m)
            matches = np.array(matches).reshape(-1, 2)
        else:
            matches = matched_indices
        ret = []
        for m in matches:
            track = results[m[0]]
            track['tracking_id'] = self.tracks[m[1]]['tracking_id']
            track['age'] = 1
            track['active'] = self.tracks[m[1]]['active'] + 1
            ret.append(track)
        if self.public_det and len(unmatched_dets) > 0:
            pub_dets = np.array([d['ct'] for d in public_det], np.float32)
            dist3 = ((dets.reshape(-1, 1, 2) - pub_dets.reshape(1, -1, 2)) ** 2).sum(axis=2)
            matched_dets = [d for d in range(dets.shape[0]) if not (d in unmatched_dets)]
            dist3[matched_dets] = 1e18
            for j in range(len(pub_dets)):
                i = dist3[:, j].argmin()
                if dist3[i, j] < item_size[i]:
                    dist3[i, :] = 1e18
                    track = results[i]
                    if track['score'] > self.new_thresh:
                        self.id_count += 1
                        track['tracking_id'] = self.id_count
                        track['age'] = 1
                        track['active'] = 1
                        ret.append(track)
        else:
            for i in unmatched_dets:
                track = results[i]
                if track['score'] > self.new_thresh:
                    self.id_count += 1
                    track['tracking_id'] = self.id_count
                    track['age'] = 1
                    track['active'] = 1
                    ret.append(track)
        for i in unmatched_tracks:
            track = self.tracks[i]
            if track['age'] < self.max_age:
                track['age'] += 1
                track['active'] = 0
                bbox = track['bbox']
                ct = track['ct']
                v = [0, 0]
                track['bbox'] = [bbox[0] + v[0], bbox[1] + v[1], bbox[2] + v[0], bbox[3] + v[1]]
                track['ct'] = [ct[0] + v[0], ct[1] + v[1]]
                ret.append(track)
        self.tracks = ret
        return ret
