"""Host-side geometry of the hot path (numpy): the crop/scale affine, its inverse,
and the Gaussian prior heat-map rendering parameters.

Mirrors the reference's interface for this path (same names / argument meaning):
``get_affine_transform`` (src/lib/utils/image.py:37-70), ``affine_transform`` (:73-76),
``transform_preds_with_trans`` (:20-26), ``gaussian_radius`` (:105-126),
``gaussian2D`` (:130-136), ``draw_umich_gaussian`` (:139-154) and the ``meta`` dict of
``Detector.pre_process`` (src/lib/detector.py:175-239).  ``cv2`` is not needed:
``cv2.getAffineTransform`` is the exact float64 solution of a 2x3 affine through three
point pairs, computed here with one 6x6 solve.
"""
import numpy as np


def _solve_affine(src, dst):
    """float64 [2,3] M with M @ (x, y, 1) = (u, v) for three (x,y)->(u,v) pairs."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    a = np.zeros((6, 6), np.float64)
    a[0:3, 0:2] = src
    a[0:3, 2] = 1.0
    a[3:6, 3:5] = src
    a[3:6, 5] = 1.0
    return np.linalg.solve(a, np.concatenate([dst[:, 0], dst[:, 1]])).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size,
                         shift=np.array([0, 0], dtype=np.float32), inv=0):
    """Same contract as the reference (image.py:37-70): three float32 anchor points
    (centre, centre + rotated (0, -src_w/2), and their 90-degree companion) mapped onto the
    output rectangle; ``inv=1`` returns the output->source transform."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    src_dir = [0 * cs - (src_w * -0.5) * sn, 0 * sn + (src_w * -0.5) * cs]
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center + scale * shift
    src[1] = center + src_dir + scale * shift
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    for pts in (src, dst):
        d = pts[0] - pts[1]
        pts[2] = pts[1] + np.array([-d[1], d[0]], dtype=np.float32)
    return _solve_affine(dst, src) if inv else _solve_affine(src, dst)


def affine_transform(pt, t):
    return np.dot(t, np.array([pt[0], pt[1], 1.], dtype=np.float32).T)[:2]


def transform_preds_with_trans(coords, trans):
    target = np.ones((coords.shape[0], 3), np.float32)
    target[:, :2] = coords
    return np.dot(trans, target.transpose()).transpose()[:, :2]


def gaussian_radius(det_size, min_overlap=0.7):
    height, width = det_size
    b1 = (height + width)
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 16 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def gaussian2D(shape, sigma=1):
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_umich_gaussian(heatmap, center, radius, k=1):
    diameter = 2 * radius + 1
    gaussian = gaussian2D((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    dst = heatmap[y - top:y + bottom, x - left:x + right]
    src = gaussian[radius - top:radius + bottom, radius - left:radius + right]
    if min(src.shape) > 0 and min(dst.shape) > 0:
        np.maximum(dst, src * k, out=dst)
    return heatmap


def make_meta(inp_height, inp_width, height, width, down_ratio=4, scale=1, calib=None,
              focal_length=1200, fix_res=True, fix_short=0, pad=31):
    """The ``meta`` dict ``Detector.pre_process`` attaches to a frame (detector.py:175-239)
    for an original image of ``height x width`` and a network input of
    ``inp_height x inp_width`` (``fix_res``), or derived sizes (``fix_short`` / keep-res)."""
    new_height, new_width = int(height * scale), int(width * scale)
    if fix_short > 0:
        if height < width:
            inp_height = fix_short
            inp_width = (int(width / height * fix_short) + 63) // 64 * 64
        else:
            inp_height = (int(height / width * fix_short) + 63) // 64 * 64
            inp_width = fix_short
        c = np.array([width / 2, height / 2], dtype=np.float32)
        s = np.array([width, height], dtype=np.float32)
    elif fix_res:
        c = np.array([new_width / 2., new_height / 2.], dtype=np.float32)
        s = max(height, width) * 1.0
    else:
        inp_height = (new_height | pad) + 1
        inp_width = (new_width | pad) + 1
        c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
        s = np.array([inp_width, inp_height], dtype=np.float32)
    out_height, out_width = inp_height // down_ratio, inp_width // down_ratio
    if calib is None:
        calib = np.array([[focal_length, 0, width / 2, 0], [0, focal_length, height / 2, 0],
                          [0, 0, 1, 0]])
    else:
        calib = np.array(calib, dtype=np.float32)
    return {'calib': calib, 'c': c, 's': s, 'height': height, 'width': width,
            'out_height': out_height, 'out_width': out_width,
            'inp_height': inp_height, 'inp_width': inp_width,
            'trans_input': get_affine_transform(c, s, 0, [inp_width, inp_height]),
            'trans_output': get_affine_transform(c, s, 0, [out_width, out_height])}
