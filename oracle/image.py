"""Oracle: host-side geometry helpers (numpy).  TEST INFRASTRUCTURE ONLY.

Restates ``src/lib/utils/image.py``: transform_preds_with_trans :20-26,
get_affine_transform :37-70, affine_transform :73-76, get_3rd_point :79-81,
get_dir :84-91, gaussian_radius :105-126, gaussian2D :130-136,
draw_umich_gaussian :139-154.  ``cv2.getAffineTransform`` (absent here) is
restated as the exact float64 solve of the 2x3 affine from 3 point pairs.
"""
import numpy as np


def get_affine_transform_3pt(src, dst):
    """cv2.getAffineTransform(src[3,2] f32, dst[3,2] f32) -> float64 [2,3]: M with
    M @ [x, y, 1]^T = [u, v]^T for the three pairs (OpenCV solves the same 6x6
    system with LU in double)."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        a[i + 3, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[i] = dst[i, 0]
        b[i + 3] = dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_affine_transform(center, scale, rot, output_size,
                         shift=np.array([0, 0], dtype=np.float32), inv=0):
    """image.py:37-70"""
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale * shift
    src[1, :] = center + src_dir + scale * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    if inv:
        return get_affine_transform_3pt(np.float32(dst), np.float32(src))
    return get_affine_transform_3pt(np.float32(src), np.float32(dst))


def affine_transform(pt, t):
    """image.py:73-76"""
    new_pt = np.array([pt[0], pt[1], 1.], dtype=np.float32).T
    return np.dot(t, new_pt)[:2]


def transform_preds_with_trans(coords, trans):
    """image.py:20-26"""
    target = np.ones((coords.shape[0], 3), np.float32)
    target[:, :2] = coords
    return np.dot(trans, target.transpose()).transpose()[:, :2]


def gaussian_radius(det_size, min_overlap=0.7):
    """image.py:105-126"""
    height, width = det_size
    a1 = 1
    b1 = (height + width)
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * a1 * c1)) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 4 * a2 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def gaussian2D(shape, sigma=1):
    """image.py:130-136"""
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_umich_gaussian(heatmap, center, radius, k=1):
    """image.py:139-154 (max-splat of a (2r+1)^2 Gaussian, sigma = (2r+1)/6)"""
    diameter = 2 * radius + 1
    gaussian = gaussian2D((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = gaussian[radius - top:radius + bottom, radius - left:radius + right]
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        np.maximum(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def warp_affine_u8(img, trans, dst_w, dst_h):
    """cv2.warpAffine(img, trans, (dst_w, dst_h), flags=INTER_LINEAR) for uint8 HxWxC images with the
    default constant border 0, restated from OpenCV's published fixed-point algorithm
    (modules/imgproc/src/imgwarp.cpp: WarpAffineInvoker + remapBilinear): inverse map in 10-bit fixed point,
    1/32-pixel coordinates, 15-bit integer tap weights, round-to-nearest.  cv2 itself is not installed here:
    PARITY UNPINNED (this vectorised numpy form is the independent checker of csrc/host_preprocess.cpp).
    Reference call site: src/lib/detector.py:218-220."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    M = np.array(trans, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11; M[0, 1] *= -D
    M[1, 0] *= -D; M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    AB = 1024
    xs = np.arange(dst_w, dtype=np.float64)
    ys = np.arange(dst_h, dtype=np.float64)
    adelta = np.rint(M[0, 0] * xs * AB).astype(np.int64)
    bdelta = np.rint(M[1, 0] * xs * AB).astype(np.int64)
    X0 = np.rint((M[0, 1] * ys + M[0, 2]) * AB).astype(np.int64) + 16
    Y0 = np.rint((M[1, 1] * ys + M[1, 2]) * AB).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    pad = np.zeros((h + 2, w + 2) + img.shape[2:], np.int64)          # zero border of one pixel
    pad[1:-1, 1:-1] = img

    def tap(yy, xx):
        ok = (yy >= -1) & (yy <= h) & (xx >= -1) & (xx <= w)
        v = pad[np.clip(yy + 1, 0, h + 1), np.clip(xx + 1, 0, w + 1)]
        return v * ok[..., None] if img.ndim == 3 else v * ok
    w00 = ((32 - fx) * (32 - fy) * 32)
    w01 = (fx * (32 - fy) * 32)
    w10 = ((32 - fx) * fy * 32)
    w11 = (fx * fy * 32)
    if img.ndim == 3:
        w00, w01, w10, w11 = w00[..., None], w01[..., None], w10[..., None], w11[..., None]
    acc = tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def pre_process_image(image, trans_input, inp_w, inp_h, mean, std, flip_test=False):
    """detector.py:218-226: warp, ((x/255 - mean)/std).astype(float32), HWC -> CHW, optional flipped copy."""
    inp = warp_affine_u8(image, trans_input, inp_w, inp_h)
    inp = ((inp / 255. - mean) / std).astype(np.float32)
    images = inp.transpose(2, 0, 1).reshape(1, 3, inp_h, inp_w)
    if flip_test:
        images = np.concatenate((images, images[:, :, :, ::-1]), axis=0)
    return images
