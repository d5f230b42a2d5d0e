"""On-disk result formats of the step after the hot path: the
MOTChallenge and KITTI-tracking text files the reference's dataset classes write before calling
their evaluators (``MOT.save_results`` datasets/mot.py:52-83, ``KITTITracking.save_results``
datasets/kitti_tracking.py:51-97).  Byte-identical output on the same results (pinned by
tests/golden/writers.json, produced by the reference's own methods); the evaluators themselves
(``tools/eval_motchallenge.py``, ``tools/eval_kitti_track``) stay the reference's.

``results`` is what ``test.py`` collects: ``{image_id: [item, ...]}`` with the items of
``Detector.run(...)['results']`` (dicts, or the structured rows of the native host path);
``videos`` / ``video_to_images`` are the tables every reference dataset holds
(``self.coco.dataset['videos']``, ``self.video_to_images``), so a maintainer's ``save_results`` can
delegate in one line::

    def save_results(self, results, save_dir):            # datasets/mot.py
        return results_io.save_mot_results(results, save_dir, self.coco.dataset['videos'],
                                           self.video_to_images, self.dataset_version)
"""
import os


def _get(item, key, default=None):
    """field of a result item: dict (reference shape) or numpy structured row (native host path)"""
    if isinstance(item, dict):
        return item.get(key, default)
    return item[key] if key in (item.dtype.names or ()) else default


def mot_lines(frames):
    """frames: iterable of (frame_id, items).  The lines of one MOTChallenge result file:
    ``frame,id,x,y,w,h,-1,-1,-1,-1`` grouped by track, tracks renumbered 1..n in ascending order of
    their tracking id, inactive items dropped (mot.py:60-83).  Width / height are formed in the
    items' own precision (float32 boxes subtract in float32, like the reference's)."""
    tracks = {}
    for frame_id, items in frames:
        for it in items:
            if int(_get(it, 'active', 1)) == 0:
                continue
            tid = _get(it, 'tracking_id')
            if tid is None:
                raise ValueError('MOT results need tracking ids (run with tracking enabled)')
            b = _get(it, 'bbox')
            tracks.setdefault(int(tid), []).append((frame_id, b[0], b[1], b[2] - b[0], b[3] - b[1]))
    lines = []
    for new_id, tid in enumerate(sorted(tracks), 1):
        for frame_id, x, y, w, h in tracks[tid]:
            lines.append('%s,%d,%.2f,%.2f,%.2f,%.2f,-1,-1,-1,-1' % (frame_id, new_id, x, y, w, h))
    return lines


def kitti_tracking_lines(frames, class_name):
    """frames: iterable of (frame_id, items).  KITTI tracking label lines (kitti_tracking.py:63-96):
    ``frame-1 track_id type -1 -1 alpha x1 y1 x2 y2 h w l x y z rot_y score`` with the reference's
    integer truncation of the 3D fields, its placeholders for absent ones (alpha -1, dim -1, loc
    -1000, rot_y -10) and its 0.01 floor on predicted dimensions."""
    lines = []
    for frame_id, items in frames:
        for it in items:
            name = class_name[int(_get(it, 'class')) - 1]
            alpha = _get(it, 'alpha', -1)
            rot_y = _get(it, 'rot_y', -10)
            dim = _get(it, 'dim')
            dim = [-1, -1, -1] if dim is None else [max(d, 0.01) for d in dim[:3]]
            loc = _get(it, 'loc')
            loc = [-1000, -1000, -1000] if loc is None else loc
            tid = _get(it, 'tracking_id', -1)
            b = _get(it, 'bbox')
            lines.append('%s %s %s -1 -1 %d %.2f %.2f %.2f %.2f %d %d %d %d %d %d %d %.2f' % (
                frame_id - 1, int(tid), name, int(alpha), b[0], b[1], b[2], b[3], int(dim[0]), int(dim[1]), int(dim[2]),
                int(loc[0]), int(loc[1]), int(loc[2]), int(rot_y), _get(it, 'score')))
    return lines


def _video_frames(results, images):
    for info in images:
        if info['id'] in results:
            yield info['frame_id'], results[info['id']]


def _write(path, lines):
    with open(path, 'w') as f:
        f.write(''.join(line + '\n' for line in lines))


def save_mot_results(results, save_dir, videos, video_to_images, dataset_version):
    """``MOT.save_results``: one ``<save_dir>/results_mot<version>/<video>.txt`` per video."""
    out_dir = os.path.join(save_dir, 'results_mot%s' % dataset_version)
    os.makedirs(out_dir, exist_ok=True)
    for video in videos:
        _write(os.path.join(out_dir, '%s.txt' % video['file_name']),
               mot_lines(_video_frames(results, video_to_images[video['id']])))
    return out_dir


def save_kitti_tracking_results(results, save_dir, videos, video_to_images,
                                class_name=('Pedestrian', 'Car', 'Cyclist')):
    """``KITTITracking.save_results``: one ``<save_dir>/results_kitti_tracking/<video>.txt`` per video."""
    out_dir = os.path.join(save_dir, 'results_kitti_tracking')
    os.makedirs(out_dir, exist_ok=True)
    for video in videos:
        _write(os.path.join(out_dir, '%s.txt' % video['file_name']),
               kitti_tracking_lines(_video_frames(results, video_to_images[video['id']]), class_name))
    return out_dir


def read_mot_results(path):
    """parse a MOTChallenge result / det file back into {frame_id: [(track_id, x, y, w, h), ...]} (round-trip
    checks and the public-detection files of ``tools/convert_mot_det_to_results.py``)"""
    frames = {}
    with open(path) as f:
        for line in f:
            p = line.strip().split(',')
            if len(p) < 6:
                continue
            frames.setdefault(int(p[0]), []).append((int(p[1]), float(p[2]), float(p[3]), float(p[4]), float(p[5])))
    return frames


def public_dets_from_mot(lines, frame_base=0):
    """MOTChallenge ``det.txt`` lines -> ``{frame_id + frame_base: [item, ...]}`` in the shape the detector takes
    as ``meta['cur_dets']`` / ``meta['pre_dets']`` for ``--public_det`` (test.py:101-107): what
    ``tools/convert_mot_det_to_results.py:31-56`` stores per image -- ``bbox`` [x1,y1,x2,y2] from the float32
    parsed x,y,w,h, ``score`` 1, ``class`` 1, ``ct`` the box centre."""
    import numpy as np
    out = {}
    for line in lines:
        p = line.strip().split(',')
        if len(p) < 6:
            continue
        x, y, w, h = (float(np.float32(v)) for v in p[2:6])
        bbox = [x, y, x + w, y + h]
        out.setdefault(int(float(p[0])) + frame_base, []).append(
            {'bbox': bbox, 'score': 1.0, 'class': 1, 'ct': [(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2]})
    return out
