"""Host-side post-processing: output-grid detections -> original-image coordinates.

Same contract as the reference ``generic_post_process`` (src/lib/utils/post_process.py:
21-91, non-pose heads) with its 3D helpers (src/lib/utils/ddd_utils.py:91-136): takes the
decode dict of numpy arrays ``[B,K,...]``, walks each image's detections in score order,
stops at the first ``score < out_thresh`` and returns per image a list of dicts
(``score, class, ct, tracking, bbox, [dep, dim, alpha, loc, rot_y, ...]``).  The float32
inverse affine and the per-detection 2x3 @ 3xn products keep the reference's shapes so the
values (and everything the tracker derives from them) are bit-identical."""
import numpy as np

from .image import get_affine_transform, transform_preds_with_trans


def get_alpha(rot):
    idx = rot[:, 1] > rot[:, 5]
    alpha1 = np.arctan2(rot[:, 2], rot[:, 3]) + (-0.5 * np.pi)
    alpha2 = np.arctan2(rot[:, 6], rot[:, 7]) + (0.5 * np.pi)
    return alpha1 * idx + alpha2 * (1 - idx)


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


def generic_post_process(opt, dets, c, s, h, w, num_classes=None, calibs=None, height=-1, width=-1):
    if 'scores' not in dets:
        return [{}], [{}]
    ret = []
    has3d = all(k in dets for k in ('rot', 'dep', 'dim'))
    for i in range(len(dets['scores'])):
        trans = get_affine_transform(c[i], s[i], 0, (w, h), inv=1).astype(np.float32)
        scores = dets['scores'][i]
        below = np.nonzero(scores < opt.out_thresh)[0]
        n = int(below[0]) if len(below) else len(scores)
        preds = []
        for j in range(n):
            ct_out = dets['cts'][i][j]
            item = {'score': scores[j], 'class': int(dets['clses'][i][j]) + 1,
                    'ct': transform_preds_with_trans(ct_out.reshape(1, 2), trans).reshape(2)}
            if 'tracking' in dets:
                moved = transform_preds_with_trans((dets['tracking'][i][j] + ct_out).reshape(1, 2), trans)
                item['tracking'] = moved.reshape(2) - item['ct']
            if 'bboxes' in dets:
                item['bbox'] = transform_preds_with_trans(dets['bboxes'][i][j].reshape(2, 2), trans).reshape(4)
            if 'dep' in dets:
                item['dep'] = dets['dep'][i][j]
            if 'dim' in dets:
                item['dim'] = dets['dim'][i][j]
            if 'rot' in dets:
                item['alpha'] = get_alpha(dets['rot'][i][j:j + 1])[0]
            if has3d:
                if 'amodel_offset' in dets:
                    ct3 = dets['bboxes'][i][j].reshape(2, 2).mean(axis=0) + dets['amodel_offset'][i][j]
                    ct = transform_preds_with_trans(ct3.reshape(1, 2), trans).reshape(2).tolist()
                else:
                    bbox = item['bbox']
                    ct = [(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2]
                item['ct'] = ct
                item['loc'], item['rot_y'] = ddd2locrot(ct, item['alpha'], item['dim'], item['dep'], calibs[i])
            preds.append(item)
        for key in ('nuscenes_att', 'velocity'):
            if key in dets:
                for j in range(len(preds)):
                    preds[j][key] = dets[key][i][j]
        ret.append(preds)
    return ret
