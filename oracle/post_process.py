"""Oracle: host-side post-processing (numpy).  TEST INFRASTRUCTURE ONLY.

Restates ``src/lib/utils/post_process.py`` (get_alpha :12-19, generic_post_process
:21-91, non-pose branches) and the 3D helpers it calls from
``src/lib/utils/ddd_utils.py`` (unproject_2d_to_3d :91-101, alpha2rot_y :103-115,
ddd2locrot :130-136).
"""
import numpy as np

from .image import get_affine_transform, transform_preds_with_trans


def get_alpha(rot):
    """post_process.py:12-19"""
    idx = rot[:, 1] > rot[:, 5]
    alpha1 = np.arctan2(rot[:, 2], rot[:, 3]) + (-0.5 * np.pi)
    alpha2 = np.arctan2(rot[:, 6], rot[:, 7]) + (0.5 * np.pi)
    return alpha1 * idx + alpha2 * (1 - idx)


def unproject_2d_to_3d(pt_2d, depth, P):
    """ddd_utils.py:91-101"""
    z = depth - P[2, 3]
    x = (pt_2d[0] * depth - P[0, 3] - P[0, 2] * z) / P[0, 0]
    y = (pt_2d[1] * depth - P[1, 3] - P[1, 2] * z) / P[1, 1]
    return np.array([x, y, z], dtype=np.float32).reshape(3)


def alpha2rot_y(alpha, x, cx, fx):
    """ddd_utils.py:103-115"""
    rot_y = alpha + np.arctan2(x - cx, fx)
    if rot_y > np.pi:
        rot_y -= 2 * np.pi
    if rot_y < -np.pi:
        rot_y += 2 * np.pi
    return rot_y


def ddd2locrot(center, alpha, dim, depth, calib):
    """ddd_utils.py:130-136"""
    locations = unproject_2d_to_3d(center, depth, calib)
    locations[1] += dim[0] / 2
    rotation_y = alpha2rot_y(alpha, center[0], calib[0, 2], calib[0, 0])
    return locations, rotation_y


def generic_post_process(out_thresh, dets, c, s, h, w, calibs=None):
    """post_process.py:21-91.  ``dets``: dict of numpy [B,K,...]; c, s, calibs: lists
    per image; (w, h) = output grid size.  Stops at the first score < out_thresh."""
    if 'scores' not in dets:
        return [{}], [{}]
    ret = []
    for i in range(len(dets['scores'])):
        preds = []
        trans = get_affine_transform(c[i], s[i], 0, (w, h), inv=1).astype(np.float32)
        for j in range(len(dets['scores'][i])):
            if dets['scores'][i][j] < out_thresh:
                break
            item = {}
            item['score'] = dets['scores'][i][j]
            item['class'] = int(dets['clses'][i][j]) + 1
            item['ct'] = transform_preds_with_trans(
                (dets['cts'][i][j]).reshape(1, 2), trans).reshape(2)
            if 'tracking' in dets:
                tracking = transform_preds_with_trans(
                    (dets['tracking'][i][j] + dets['cts'][i][j]).reshape(1, 2), trans).reshape(2)
                item['tracking'] = tracking - item['ct']
            if 'bboxes' in dets:
                item['bbox'] = transform_preds_with_trans(
                    dets['bboxes'][i][j].reshape(2, 2), trans).reshape(4)
            if 'dep' in dets and len(dets['dep'][i]) > j:
                item['dep'] = dets['dep'][i][j]
            if 'dim' in dets and len(dets['dim'][i]) > j:
                item['dim'] = dets['dim'][i][j]
            if 'rot' in dets and len(dets['rot'][i]) > j:
                item['alpha'] = get_alpha(dets['rot'][i][j:j + 1])[0]
            if 'rot' in dets and 'dep' in dets and 'dim' in dets and len(dets['dep'][i]) > j:
                if 'amodel_offset' in dets and len(dets['amodel_offset'][i]) > j:
                    ct_output = dets['bboxes'][i][j].reshape(2, 2).mean(axis=0)
                    amodel_ct_output = ct_output + dets['amodel_offset'][i][j]
                    ct = transform_preds_with_trans(
                        amodel_ct_output.reshape(1, 2), trans).reshape(2).tolist()
                else:
                    bbox = item['bbox']
                    ct = [(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2]
                item['ct'] = ct
                item['loc'], item['rot_y'] = ddd2locrot(
                    ct, item['alpha'], item['dim'], item['dep'], calibs[i])
            preds.append(item)
        if 'nuscenes_att' in dets:
            for j in range(len(preds)):
                preds[j]['nuscenes_att'] = dets['nuscenes_att'][i][j]
        if 'velocity' in dets:
            for j in range(len(preds)):
                preds[j]['velocity'] = dets['velocity'][i][j]
        ret.append(preds)
    return ret
