"""Host-side post-processing: decoded detections on the output grid -> original-image coordinates.

Python form of the host step the native path runs in C++ (``ct_tracker_step``); it serves the branches the
native tracker does not cover (``--hungarian``, ``--public_det``, ``pre_dets``) and every 3D field.  Same
contract as the reference's ``generic_post_process`` (src/lib/utils/post_process.py:21-91)
with the 3D helpers of src/lib/utils/ddd_utils.py:91-136: input = the decode dict of numpy arrays
``[B,K,...]``; output = per image a list of dicts (``score, class, ct, tracking, bbox, [hps, dep, dim, alpha,
loc, rot_y, nuscenes_att, velocity]``) for the detections ahead of the first ``score < out_thresh``.
Every point goes through the same float32 ``[2,3] @ [3,1]`` product as in the reference, so the values (and
what the tracker derives from them) are bit-identical.
"""
import numpy as np

from .image import get_affine_transform, transform_preds_with_trans


class _GridToImage(object):
    """float32 inverse affine of one image (output grid -> original image), post_process.py:30"""

    def __init__(self, c, s, out_w, out_h):
        self.m = get_affine_transform(c, s, 0, (out_w, out_h), inv=1).astype(np.float32)

    def point(self, xy):
        return transform_preds_with_trans(np.asarray(xy).reshape(1, 2), self.m).reshape(2)

    def box(self, x0y0x1y1):
        return transform_preds_with_trans(np.asarray(x0y0x1y1).reshape(2, 2), self.m).reshape(4)

    def points(self, flat_xy):
        """[x0,y0,x1,y1,...] -> same layout in image coordinates (key points, post_process.py:51-54)"""
        return transform_preds_with_trans(np.asarray(flat_xy).reshape(-1, 2), self.m).reshape(-1)


def get_alpha(rot):
    """observation angle from the 8-bin rotation vector (post_process.py:12-19)"""
    use_first = rot[:, 1] > rot[:, 5]
    from_first = np.arctan2(rot[:, 2], rot[:, 3]) + (-0.5 * np.pi)
    from_second = np.arctan2(rot[:, 6], rot[:, 7]) + (0.5 * np.pi)
    return from_first * use_first + from_second * (1 - use_first)


def unproject_2d_to_3d(pt_2d, depth, P):
    z = depth - P[2, 3]
    x = (pt_2d[0] * depth - P[0, 3] - P[0, 2] * z) / P[0, 0]
    y = (pt_2d[1] * depth - P[1, 3] - P[1, 2] * z) / P[1, 1]
    return np.array([x, y, z], dtype=np.float32).reshape(3)


def alpha2rot_y(alpha, x, cx, fx):
    rot_y = alpha + np.arctan2(x - cx, fx)
    if rot_y > np.pi:
        rot_y -= 2 * np.pi
    if rot_y < -np.pi:
        rot_y += 2 * np.pi
    return rot_y


def ddd2locrot(center, alpha, dim, depth, calib):
    loc = unproject_2d_to_3d(center, depth, calib)
    loc[1] += dim[0] / 2
    return loc, alpha2rot_y(alpha, center[0], calib[0, 2], calib[0, 0])


def _lift_to_3d(item, fields, j, to_img, calib):
    """3D branch (post_process.py:59-80): amodal centre -> location / yaw"""
    if 'amodel_offset' in fields:
        grid_ct = fields['bboxes'][j].reshape(2, 2).mean(axis=0) + fields['amodel_offset'][j]
        ct = to_img.point(grid_ct).tolist()
    else:
        b = item['bbox']
        ct = [(b[0] + b[2]) / 2, (b[1] + b[3]) / 2]
    item['ct'] = ct
    item['loc'], item['rot_y'] = ddd2locrot(ct, item['alpha'], item['dim'], item['dep'], calib)


def generic_post_process(opt, dets, c, s, h, w, num_classes=None, calibs=None, height=-1, width=-1):
    if 'scores' not in dets:
        return [{}], [{}]
    want_3d = all(k in dets for k in ('rot', 'dep', 'dim'))
    per_image = []
    for i, scores in enumerate(dets['scores']):
        to_img = _GridToImage(c[i], s[i], w, h)
        fields = {k: v[i] for k, v in dets.items()}
        low = np.nonzero(scores < opt.out_thresh)[0]
        count = int(low[0]) if len(low) else len(scores)
        found = []
        for j in range(count):
            grid_ct = fields['cts'][j]
            item = {'score': scores[j], 'class': int(fields['clses'][j]) + 1, 'ct': to_img.point(grid_ct)}
            if 'tracking' in fields:
                item['tracking'] = to_img.point(fields['tracking'][j] + grid_ct) - item['ct']
            if 'bboxes' in fields:
                item['bbox'] = to_img.box(fields['bboxes'][j])
            if 'hps' in fields:
                item['hps'] = to_img.points(fields['hps'][j])
            for k in ('dep', 'dim'):
                if k in fields:
                    item[k] = fields[k][j]
            if 'rot' in fields:
                item['alpha'] = get_alpha(fields['rot'][j:j + 1])[0]
            if want_3d:
                _lift_to_3d(item, fields, j, to_img, calibs[i])
            found.append(item)
        for k in ('nuscenes_att', 'velocity'):
            if k in fields:
                for j, item in enumerate(found):
                    item[k] = fields[k][j]
        per_image.append(found)
    return per_image
