"""Native host path of one frame (ctypes over the CPU entry points of libcentertrack_hip):
packed decode rows -> post-process -> threshold -> greedy association, and the prior
heat-map blob parameters of the next frame.  Same semantics as ``post_process.py`` +
``tracker.Tracker`` (the reference-shaped Python path, kept for the Hungarian /
public-detection / pre_dets branches); results come back as a structured numpy array and are
turned into the reference's list-of-dicts only on request."""
import ctypes

import numpy as np

from . import _lib

TRACK_DTYPE = np.dtype([('score', np.float32), ('class', np.int32), ('ct', np.float32, 2),
                        ('tracking', np.float32, 2), ('bbox', np.float32, 4), ('tracking_id', np.int32),
                        ('age', np.int32), ('active', np.int32), ('row', np.int32)], align=True)
assert TRACK_DTYPE.itemsize == ctypes.sizeof(_lib.Track)
MAX_BLOBS = 256


def row_layout(layout):
    """ops.decode_layout() -> ct_row_layout"""
    off = {name: s for name, s, _ in layout}
    lay = _lib.RowLayout()
    lay.score, lay.cls, lay.cts = off['scores'], off['clses'], off['xs']
    lay.tracking = off.get('tracking', -1)
    lay.bbox = off.get('bboxes', -1)
    lay.amodel_offset = off.get('amodel_offset', -1)
    return lay


class FastTracker(object):
    """One stream's tracker state in native code."""

    def __init__(self, new_thresh, max_age, K):
        self.lib = _lib.load()
        self.h = ctypes.c_void_p(self.lib.ct_tracker_create(float(new_thresh), int(max_age)))
        self.cap = 2 * K + 64
        self.buf = np.zeros(self.cap, TRACK_DTYPE)
        self._buf_ptr = self.buf.ctypes.data
        self.params = np.zeros((MAX_BLOBS, 3), np.int32)

    def __del__(self):
        try:
            self.lib.ct_tracker_destroy(self.h)
        except Exception:
            pass

    def reset(self):
        self.lib.ct_tracker_reset(self.h)

    @property
    def id_count(self):
        return self.lib.ct_tracker_id_count(self.h)

    def step(self, rows, lay, out_thresh, trans_inv):
        """rows: C-contiguous float32 [K,F] (host); trans_inv: float32 [2,3].  Returns a view
        of the structured result array (valid until the next call)."""
        K, F = rows.shape
        n = self.lib.ct_tracker_step(self.h, rows.ctypes.data, K, F, ctypes.byref(lay), float(out_thresh),
                                     trans_inv.ctypes.data, self._buf_ptr, self.cap)
        if n < 0:
            _lib.check(1, 'ct_tracker_step')
        return self.buf[:n]

    @property
    def tracks(self):
        """current track list as a structured array (copy)"""
        tmp = np.zeros(self.cap, TRACK_DTYPE)
        n = self.lib.ct_tracker_get_tracks(self.h, tmp.ctypes.data, self.cap)
        return tmp[:min(n, self.cap)]

    def prehm_params(self, pre_thresh, trans_input, inp_w, inp_h, out=None):
        """(cx, cy, radius) of every active track with score >= pre_thresh -> (n, int32 [n,3])"""
        dst = self.params if out is None else out
        t = np.ascontiguousarray(trans_input, np.float64)
        n = self.lib.ct_tracker_prehm_params(self.h, float(pre_thresh), t.ctypes.data, int(inp_w), int(inp_h),
                                             dst.ctypes.data, dst.shape[0])
        if n < 0:
            _lib.check(1, 'ct_tracker_prehm_params')
        return n, dst


def as_dicts(arr, dets=None, stream=0):
    """structured result array -> the reference's list of dicts (extra decode fields such as
    dep / dim / rot are attached from ``dets`` via the source row when given)."""
    out = []
    for r in arr:
        d = {'score': r['score'], 'class': int(r['class']), 'ct': r['ct'].copy(), 'tracking': r['tracking'].copy(),
             'bbox': r['bbox'].copy(), 'tracking_id': int(r['tracking_id']), 'age': int(r['age']),
             'active': int(r['active'])}
        if dets is not None and r['row'] >= 0:
            for k in ('dep', 'dim', 'rot', 'nuscenes_att', 'velocity'):
                if k in dets:
                    d[k] = dets[k][stream][r['row']]
        out.append(d)
    return out
