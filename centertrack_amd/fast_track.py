"""Native host path of one frame (ctypes over the CPU entry points of libcentertrack_hip):
packed decode rows -> post-process -> threshold -> association (greedy, or Hungarian /
public-detection mode), and the prior heat-map blob parameters of the next frame.  Same
semantics as ``post_process.py`` + ``tracker.Tracker`` (the reference-shaped Python path);
results come back as a structured numpy array and are turned into the reference's
list-of-dicts only on request."""
import ctypes

import numpy as np

from . import _lib

TRACK_DTYPE = np.dtype([('score', np.float32), ('class', np.int32), ('ct', np.float32, 2),
                        ('tracking', np.float32, 2), ('bbox', np.float32, 4), ('tracking_id', np.int32),
                        ('age', np.int32), ('active', np.int32), ('row', np.int32)], align=True)
assert TRACK_DTYPE.itemsize == ctypes.sizeof(_lib.Track)
MAX_BLOBS = 256        # minimum capacity of a stream's prior-heat-map blob table (grown to K when K is larger)


def max_blobs(K):
    """blob-table rows per stream: active tracks with score >= pre_thresh are this frame's detections, <= K"""
    return max(MAX_BLOBS, int(K))


def row_layout(layout):
    """ops.decode_layout() -> ct_row_layout"""
    off = {name: s for name, s, _ in layout}
    lay = _lib.RowLayout()
    lay.score, lay.cls, lay.cts = off['scores'], off['clses'], off['xs']
    lay.tracking = off.get('tracking', -1)
    lay.bbox = off.get('bboxes', -1)
    lay.amodel_offset = off.get('amodel_offset', -1)
    return lay


def items_to_array(items):
    """list of result dicts -> TRACK_DTYPE array (missing 'ct' = box centre, missing 'tracking' = 0)"""
    arr = np.zeros(len(items), TRACK_DTYPE)
    for i, it in enumerate(items):
        arr[i]['score'] = it['score']
        arr[i]['class'] = it.get('class', 1)
        arr[i]['bbox'] = np.asarray(it['bbox'], np.float32)
        if 'ct' in it:
            arr[i]['ct'] = np.asarray(it['ct'], np.float32)
        else:
            b = it['bbox']
            arr[i]['ct'] = [(b[0] + b[2]) / 2, (b[1] + b[3]) / 2]
        if 'tracking' in it:
            arr[i]['tracking'] = np.asarray(it['tracking'], np.float32)
        arr[i]['row'] = -1
    return arr


def _public_centres(public_det):
    if public_det is None or len(public_det) == 0:
        return np.zeros((0, 2), np.float32)
    return np.ascontiguousarray([d['ct'] for d in public_det], np.float32).reshape(-1, 2)


class FastTracker(object):
    """One stream's tracker state in native code."""

    def __init__(self, new_thresh, max_age, K, hungarian=False, public_det=False):
        self.lib = _lib.load()
        self.h = ctypes.c_void_p(self.lib.ct_tracker_create(float(new_thresh), int(max_age)))
        self.public_det = bool(public_det)
        if hungarian or public_det:
            _lib.check(self.lib.ct_tracker_set_mode(self.h, int(bool(hungarian)), int(bool(public_det))),
                       'ct_tracker_set_mode')
        self.cap = 2 * K + 64
        self.buf = np.zeros(self.cap, TRACK_DTYPE)
        self._buf_ptr = self.buf.ctypes.data
        self.params = np.zeros((max_blobs(K), 3), np.int32)

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

    def step(self, rows, lay, out_thresh, trans_inv, public_det=None):
        """rows: C-contiguous float32 [K,F] (host); trans_inv: float32 [2,3]; public_det: the frame's provided
        detections (list of dicts with 'ct', tracker.py:85) in public-detection mode.  Returns a view of the
        structured result array (valid until the next call)."""
        K, F = rows.shape
        if self.public_det:
            pub = _public_centres(public_det)
            n = self.lib.ct_tracker_step_public(self.h, rows.ctypes.data, K, F, ctypes.byref(lay), float(out_thresh),
                                                trans_inv.ctypes.data, pub.ctypes.data, len(pub), self._buf_ptr,
                                                self.cap)
        else:
            n = self.lib.ct_tracker_step(self.h, rows.ctypes.data, K, F, ctypes.byref(lay), float(out_thresh),
                                         trans_inv.ctypes.data, self._buf_ptr, self.cap)
        return self._result(n, 'ct_tracker_step')

    def _result(self, n, what):
        """the step's tracks; the list is unbounded in the reference (max_age > 0 accumulates unmatched tracks), so a
        result larger than the buffer grows it and is re-read from the tracker (the step itself always completes)"""
        if n < 0:
            _lib.check(1, what)
        if n > self.cap:
            self.cap = 2 * n
            self.buf = np.zeros(self.cap, TRACK_DTYPE)
            self._buf_ptr = self.buf.ctypes.data
            got = self.lib.ct_tracker_get_tracks(self.h, self._buf_ptr, self.cap)
            assert got == n, (got, n)
        return self.buf[:n]

    def step_dets(self, results, public_det=None):
        """``Tracker.step(results, public_det)`` (tracker.py:28) on image-space detections: a list of dicts
        (score, class, ct, tracking, bbox) or a TRACK_DTYPE array"""
        dets = results if isinstance(results, np.ndarray) else items_to_array(results)
        pub = _public_centres(public_det) if self.public_det else np.zeros((0, 2), np.float32)
        n = self.lib.ct_tracker_step_dets(self.h, dets.ctypes.data, len(dets), pub.ctypes.data, len(pub),
                                          self._buf_ptr, self.cap)
        return self._result(n, 'ct_tracker_step_dets')

    def init_tracks(self, results):
        """``Tracker.init_track(results)`` (tracker.py:13-26): detections with score > new_thresh start tracks;
        an item without 'ct' takes its box centre"""
        dets = items_to_array(results)
        if self.lib.ct_tracker_init_tracks(self.h, dets.ctypes.data, len(dets)) < 0:
            _lib.check(1, 'ct_tracker_init_tracks')

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


def as_dicts(arr, dets=None, stream=0, calib=None, carried=None, trans_inv=None):
    """structured result array -> the reference's list of dicts.  Fields the native rows do not carry are attached
    from ``dets`` (the decode dict of the frame) via the source row, like generic_post_process does
    (post_process.py:56-88): dep, dim, the observation angle ``alpha`` from the 8-bin ``rot``, and with ``calib``
    the 3D location / yaw (``loc``, ``rot_y``; the amodal centre is already the row's ``ct``), nuscenes_att,
    velocity; with ``trans_inv`` (the float32 [2,3] output-grid -> image affine of the frame) the key points ``hps``
    of the pose task in image coordinates (post_process.py:51-54, native ``ct_transform_points``).  ``carried``: optional dict {tracking_id: extras} owned by the caller; tracks kept alive without a
    detection (row < 0, ``max_age``) get back the extras of their last detection, as the reference's track dicts do."""
    from .post_process import ddd2locrot, get_alpha
    out = []
    alive = set()
    for r in arr:
        d = {'score': r['score'], 'class': int(r['class']), 'ct': r['ct'].copy(), 'tracking': r['tracking'].copy(),
             'bbox': r['bbox'].copy(), 'tracking_id': int(r['tracking_id']), 'age': int(r['age']),
             'active': int(r['active'])}
        row = int(r['row'])
        if dets is not None and row >= 0:
            extras = {}
            for k in ('dep', 'dim'):
                if k in dets:
                    extras[k] = np.array(dets[k][stream][row])     # (own copy: `dets` may view a reused buffer)
            if 'rot' in dets:
                extras['alpha'] = get_alpha(dets['rot'][stream][row:row + 1])[0]
            if calib is not None and all(k in dets for k in ('rot', 'dep', 'dim')):
                ct = d['ct'].tolist()                        # (ddd: the projected amodal / box centre, post_process.py:66-80)
                d['ct'] = ct
                extras['loc'], extras['rot_y'] = ddd2locrot(ct, extras['alpha'], extras['dim'], extras['dep'], calib)
            for k in ('nuscenes_att', 'velocity'):
                if k in dets:
                    extras[k] = np.array(dets[k][stream][row])
            if 'hps' in dets and trans_inv is not None:
                pts = np.ascontiguousarray(dets['hps'][stream][row], np.float32)
                out_pts = np.empty_like(pts)
                if _lib.load().ct_transform_points(trans_inv.ctypes.data, pts.ctypes.data, pts.size // 2,
                                                   out_pts.ctypes.data) < 0:
                    _lib.check(1, 'ct_transform_points')
                extras['hps'] = out_pts
            d.update(extras)
            if carried is not None:
                carried[d['tracking_id']] = extras
        elif carried is not None and d['tracking_id'] in carried:
            d.update(carried[d['tracking_id']])
        alive.add(d['tracking_id'])
        out.append(d)
    if carried is not None:
        for tid in [t for t in carried if t not in alive]:
            del carried[tid]
    return out
