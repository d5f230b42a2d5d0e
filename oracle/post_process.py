"""Oracle: host-side post-processing of the decoded detections (numpy).

TEST INFRASTRUCTURE ONLY.  Restates the behaviour of ``generic_post_process`` / ``get_alpha``
(src/lib/utils/post_process.py:12-91, non-pose heads) and of the 3D helpers it calls
(src/lib/utils/ddd_utils.py: ``unproject_2d_to_3d`` :91-101, ``alpha2rot_y`` :103-115, ``ddd2locrot``
:130-136).  Pinned against the imported reference by tests/golden/make_golden.py (post_process.json).

What the reference does, per image: walk the K detections in score order, stop at the first score below
``out_thresh``, and map centre, displaced centre and box corners from the output grid to the original image
with the float32 inverse affine; for 3D heads pick the observation angle from the 8-bin ``rot`` vector, move
the centre to the projected amodal centre and un-project it with the calibration matrix.
"""
import numpy as np

from .image import get_affine_transform, transform_preds_with_trans

_HALF_PI = 0.5 * np.pi


def get_alpha(rot):
    """[n,8] = two bins of (cls0, cls1, sin, cos): the bin whose cls1 wins gives atan2(sin, cos) -/+ pi/2"""
    first_bin = rot[:, 1] > rot[:, 5]
    a_first = np.arctan2(rot[:, 2], rot[:, 3]) + (-_HALF_PI)
    a_second = np.arctan2(rot[:, 6], rot[:, 7]) + _HALF_PI
    return a_first * first_bin + a_second * (1 - first_bin)


def unproject_2d_to_3d(pt_2d, depth, P):
    z = depth - P[2, 3]
    xy = [(pt_2d[k] * depth - P[k, 3] - P[k, 2] * z) / P[k, k] for k in (0, 1)]
    return np.array([xy[0], xy[1], z], dtype=np.float32).reshape(3)


def alpha2rot_y(alpha, x, cx, fx):
    yaw = alpha + np.arctan2(x - cx, fx)
    if yaw > np.pi:
        yaw -= 2 * np.pi
    if yaw < -np.pi:
        yaw += 2 * np.pi
    return yaw


def ddd2locrot(center, alpha, dim, depth, calib):
    loc = unproject_2d_to_3d(center, depth, calib)
    loc[1] += dim[0] / 2                                   # box centre -> bottom centre (KITTI convention)
    return loc, alpha2rot_y(alpha, center[0], calib[0, 2], calib[0, 0])


def _to_image(points, trans):
    return transform_preds_with_trans(points, trans)


def _one_detection(dets, i, j, trans, calib):
    """the result dict of detection j of image i"""
    has = lambda k: k in dets and len(dets[k][i]) > j
    ct_grid = dets['cts'][i][j]
    item = {'score': dets['scores'][i][j], 'class': int(dets['clses'][i][j]) + 1,
            'ct': _to_image(ct_grid.reshape(1, 2), trans).reshape(2)}
    if 'tracking' in dets:
        moved = _to_image((dets['tracking'][i][j] + ct_grid).reshape(1, 2), trans).reshape(2)
        item['tracking'] = moved - item['ct']
    if 'bboxes' in dets:
        item['bbox'] = _to_image(dets['bboxes'][i][j].reshape(2, 2), trans).reshape(4)
    if 'hps' in dets:                                                  # post_process.py:51-54
        item['hps'] = _to_image(dets['hps'][i][j].reshape(-1, 2), trans).reshape(-1)
    for k in ('dep', 'dim'):
        if has(k):
            item[k] = dets[k][i][j]
    if has('rot'):
        item['alpha'] = get_alpha(dets['rot'][i][j:j + 1])[0]
    if 'rot' in dets and 'dim' in dets and has('dep'):
        if has('amodel_offset'):
            centre_grid = dets['bboxes'][i][j].reshape(2, 2).mean(axis=0) + dets['amodel_offset'][i][j]
            centre = _to_image(centre_grid.reshape(1, 2), trans).reshape(2).tolist()
        else:
            x0, y0, x1, y1 = item['bbox']
            centre = [(x0 + x1) / 2, (y0 + y1) / 2]
        item['ct'] = centre
        item['loc'], item['rot_y'] = ddd2locrot(centre, item['alpha'], item['dim'], item['dep'], calib)
    return item


def generic_post_process(out_thresh, dets, c, s, h, w, calibs=None):
    """``dets``: dict of numpy [B,K,...]; c, s, calibs: per-image lists; (w, h): output grid size."""
    if 'scores' not in dets:
        return [{}], [{}]
    images = []
    for i, scores in enumerate(dets['scores']):
        trans = get_affine_transform(c[i], s[i], 0, (w, h), inv=1).astype(np.float32)
        calib = calibs[i] if calibs is not None else None
        preds = []
        for j, score in enumerate(scores):
            if score < out_thresh:
                break
            preds.append(_one_detection(dets, i, j, trans, calib))
        for extra in ('nuscenes_att', 'velocity'):
            if extra in dets:
                for j, p in enumerate(preds):
                    p[extra] = dets[extra][i][j]
        images.append(preds)
    return images
