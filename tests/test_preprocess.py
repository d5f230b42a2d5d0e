"""Host pre-processing (Detector.pre_process, src/lib/detector.py:207-239): the C++ restatement of
cv2.warpAffine + normalisation behind ct_preprocess_image against the independent numpy restatement in
oracle/image.py, plus known-answer cases.  cv2 is not installed: parity with cv2 itself is unpinned."""
import ctypes
import types

import numpy as np
import pytest

from centertrack_amd import _lib
from centertrack_amd.detector import MEAN, STD
from centertrack_amd.image import get_affine_transform, make_meta
from oracle import image as oimage


def _run(img, trans, dw, dh, flip=False):
    lib = _lib.load()
    img = np.ascontiguousarray(img)
    ch = img.shape[2]
    out = np.empty((2 if flip else 1, ch, dh, dw), np.float32)
    trans = np.ascontiguousarray(trans, np.float64)
    mean, std = np.ascontiguousarray(MEAN.reshape(-1)), np.ascontiguousarray(STD.reshape(-1))
    rc = lib.ct_preprocess_image(img.ctypes.data_as(ctypes.c_void_p), img.shape[0], img.shape[1], img.strides[0], ch,
                                 trans.ctypes.data_as(ctypes.c_void_p), dw, dh, mean.ctypes.data_as(ctypes.c_void_p),
                                 std.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), int(flip))
    assert rc == 0, lib.ct_last_error()
    return out


def _norm(u8):
    return ((u8 / 255. - MEAN) / STD).astype(np.float32).transpose(2, 0, 1)[None]


def test_identity_and_integer_shift_are_exact():
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    ident = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    np.testing.assert_array_equal(_run(img, ident, 53, 37), _norm(img))
    shift = np.array([[1, 0, 5], [0, 1, -3]], np.float64)          # dst(x, y) = src(x - 5, y + 3), zero outside
    want = np.zeros_like(img)
    want[:37 - 3, 5:] = img[3:, :53 - 5]
    np.testing.assert_array_equal(_run(img, shift, 53, 37), _norm(want))


def test_half_pixel_blend_rounds_to_nearest():
    img = np.zeros((4, 6, 3), np.uint8)
    img[:, ::2] = 10
    img[:, 1::2] = 13
    half = np.array([[1, 0, 0.5], [0, 1, 0]], np.float64)          # samples at x - 0.5: (a + b) / 2 -> 11.5 -> 12
    got = _run(img, half, 6, 4)
    want = np.full((4, 6, 3), 12, np.uint8)
    want[:, 0] = 5                                                   # half of pixel 0 + half of the zero border
    np.testing.assert_array_equal(got, _norm(want))


@pytest.mark.parametrize('h,w,inp_h,inp_w,flip', [(360, 480, 128, 160, False), (375, 1242, 96, 320, True),
                                                   (120, 90, 64, 64, False), (33, 47, 64, 96, True)])
def test_reference_crop_matches_numpy_restatement(h, w, inp_h, inp_w, flip):
    rs = np.random.RandomState(h + w)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    meta = make_meta(inp_h, inp_w, h, w)
    got = _run(img, meta['trans_input'], inp_w, inp_h, flip)
    want = oimage.pre_process_image(img, meta['trans_input'], inp_w, inp_h, MEAN, STD, flip)
    np.testing.assert_array_equal(got, want)
    # rotated / sheared maps exercise negative coordinates and all four border cases
    t = get_affine_transform(np.array([w / 2., h / 2.], np.float32), max(h, w) * 0.7, 25, [inp_w, inp_h])
    np.testing.assert_array_equal(_run(img, t, inp_w, inp_h), oimage.pre_process_image(img, t, inp_w, inp_h, MEAN, STD))


def test_detector_pre_process_returns_the_reference_contract():
    """Detector.pre_process -> (images [1|2,3,H,W] float32 tensor, meta with the reference's keys); the class is
    not instantiated (that needs a GPU): the method only reads opt / mean / std."""
    import torch
    from centertrack_amd.detector import Detector
    fake = types.SimpleNamespace(opt=types.SimpleNamespace(input_h=128, input_w=160, down_ratio=4, fix_res=True,
                                                           fix_short=0, pad=31, flip_test=True),
                                 mean=MEAN, std=STD, rest_focal_length=1200)
    fake.frame_meta = lambda image, im={}: Detector.frame_meta(fake, image, im)
    img = np.random.RandomState(3).randint(0, 256, (360, 480, 3)).astype(np.uint8)
    images, meta = Detector.pre_process(fake, img, 1.0, {'pre_dets': []})
    assert isinstance(images, torch.Tensor) and images.dtype == torch.float32 and tuple(images.shape) == (2, 3, 128, 160)
    assert torch.equal(images[1], torch.flip(images[0], [2]))
    for k in ('calib', 'c', 's', 'height', 'width', 'out_height', 'out_width', 'inp_height', 'inp_width', 'trans_input',
              'trans_output', 'pre_dets'):
        assert k in meta
    want = oimage.pre_process_image(img, meta['trans_input'], 160, 128, MEAN, STD, True)
    np.testing.assert_array_equal(images.numpy(), want)


def test_normalisation_table_matches_the_float64_expression():
    """ct_preprocess_lut (host; feeds the device kernel): lut[c][v] == float32((v / 255. - mean) / std)"""
    lib = _lib.load()
    lut = np.empty((3, 256), np.float32)
    mean, std = np.ascontiguousarray(MEAN.reshape(-1)), np.ascontiguousarray(STD.reshape(-1))
    assert lib.ct_preprocess_lut(mean.ctypes.data, std.ctypes.data, 3, lut.ctypes.data) == 0
    v = np.arange(256, dtype=np.uint8).reshape(256, 1, 1).repeat(3, 2)          # "image" of 256 x 1 pixels
    want = ((v / 255. - MEAN) / STD).astype(np.float32)[:, 0, :].T
    np.testing.assert_array_equal(lut, want)
    assert lib.ct_preprocess_lut(mean.ctypes.data, std.ctypes.data, 9, lut.ctypes.data) != 0


@pytest.mark.parametrize('mode', ['fix_res', 'fix_short', 'keep_res'])
@pytest.mark.parametrize('h,w', [(375, 1242), (1080, 1920), (640, 480)])
def test_meta_of_every_testing_mode_equals_the_oracle(mode, h, w):
    """detector.py:175-239: fixed resolution / fixed short side / keep resolution padded to 32 -- sizes, centre,
    scale, both affine maps and the default calibration of image.make_meta == oracle/detector.make_meta, and the host
    pre-processing of that mode == the numpy restatement"""
    from oracle import detector as odet
    kw = dict(fix_res=(mode == 'fix_res'), fix_short=(256 if mode == 'fix_short' else 0), pad=31)
    got = make_meta(128, 160, h, w, **kw)
    oopt = odet.default_opt(input_h=128, input_w=160, **kw)
    want = odet.make_meta(oopt, h, w)
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_array_equal(np.asarray(got[k]), np.asarray(want[k]), err_msg=k)
    if mode == 'fix_short':
        assert min(got['inp_height'], got['inp_width']) == 256 and max(got['inp_height'], got['inp_width']) % 64 == 0
    if mode == 'keep_res':
        assert got['inp_height'] % 32 == 0 and got['inp_height'] >= h and got['inp_width'] >= w
    if h * w <= 480 * 640:
        img = np.random.RandomState(h).randint(0, 256, (h, w, 3)).astype(np.uint8)
        np.testing.assert_array_equal(
            _run(img, got['trans_input'], got['inp_width'], got['inp_height']),
            oimage.pre_process_image(img, got['trans_input'], got['inp_width'], got['inp_height'], MEAN, STD))


@pytest.mark.parametrize('h,w,inp_h,inp_w', [(360, 480, 128, 160), (375, 1242, 96, 320), (64, 64, 96, 96)])
def test_warp_stays_within_the_fixed_point_bound_of_an_independent_bilinear_sampler(h, w, inp_h, inp_w):
    """cv2 is absent, so the fixed-point warp is PARITY UNPINNED against OpenCV itself; this bounds it with an
    independently maintained sampler instead: ``F.grid_sample`` (float64, zeros padding) evaluated at the exact
    inverse-affine coordinates is the ideal bilinear warp, and OpenCV's published scheme differs from it only by
    its quantisation -- coordinates rounded to 1/32 px (<= 1/64 px error per axis, i.e. <= 255/64 grey levels each on
    a unit-step edge), 15-bit weights, round-to-nearest: |ours - ideal| <= 2*255/64 + 1 everywhere, and the mean
    difference on a smooth image stays below half a grey level.  Catches a wrong inverse, a half-pixel shift, a
    transposed matrix or a border slip -- none of which the bit-for-bit host == numpy == device checks can see
    if all three shared the misreading."""
    import torch
    import torch.nn.functional as F
    from centertrack_amd import image as IM
    from oracle import image as oimage
    rs = np.random.RandomState(h + w)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = 127 + 100 * np.sin(xx / 9.0 + rs.uniform(0, 3)) * np.cos(yy / 7.0)
    img = np.clip(np.stack([smooth, smooth[::-1], smooth[:, ::-1]], -1) + rs.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)
    meta = IM.make_meta(inp_h, inp_w, h, w)
    for trans in (meta['trans_input'],
                  IM.get_affine_transform(np.array([w / 3.0, h / 2.0], np.float32), max(h, w) * 0.6, 0, [inp_w, inp_h])):
        got = oimage.warp_affine_u8(img, trans, inp_w, inp_h).astype(np.float64)
        Minv = np.linalg.inv(np.vstack([np.asarray(trans, np.float64), [0, 0, 1]]))[:2]    # dst pixel -> src coordinate
        dy, dx = np.mgrid[0:inp_h, 0:inp_w].astype(np.float64)
        sx = Minv[0, 0] * dx + Minv[0, 1] * dy + Minv[0, 2]
        sy = Minv[1, 0] * dx + Minv[1, 1] * dy + Minv[1, 2]
        grid = torch.from_numpy(np.stack([2 * sx / (w - 1) - 1, 2 * sy / (h - 1) - 1], -1))[None]
        src = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
        ideal = F.grid_sample(src, grid, mode='bilinear', padding_mode='zeros', align_corners=True)[0].permute(1, 2, 0).numpy()
        err = np.abs(got - ideal)
        # the local slope bounds the effect of the 1/64 px coordinate rounding: allow (slope_x + slope_y)/64 + 1
        padded = np.pad(img.astype(np.float64), ((1, 1), (1, 1), (0, 0)))          # (the step into the zero border counts)
        gx = np.abs(np.diff(padded, axis=1)).max()
        gy = np.abs(np.diff(padded, axis=0)).max()
        assert err.max() <= (gx + gy) / 64.0 + 1.0, err.max()
        assert err.mean() < 0.5, err.mean()


def test_imread_bgr_is_cv2_imread_for_lossless_files(tmp_path):
    """Detector.run(path) (detector.py:65-66): the file decodes to the uint8 BGR array cv2.imread returns"""
    from PIL import Image
    from centertrack_amd.detector import imread_bgr
    bgr = np.random.RandomState(3).randint(0, 256, (37, 53, 3)).astype(np.uint8)
    for name in ('f.png', 'f.bmp'):
        Image.fromarray(bgr[:, :, ::-1]).save(tmp_path / name)
        got = imread_bgr(str(tmp_path / name))
        assert got.dtype == np.uint8 and got.flags['C_CONTIGUOUS']
        np.testing.assert_array_equal(got, bgr)
    gray = np.random.RandomState(4).randint(0, 256, (9, 11)).astype(np.uint8)
    Image.fromarray(gray).save(tmp_path / 'g.png')                 # (cv2.imread's default flag also yields 3 channels)
    np.testing.assert_array_equal(imread_bgr(tmp_path / 'g.png'), np.repeat(gray[:, :, None], 3, 2))


@pytest.mark.parametrize('h,w,inp_h,inp_w', [(360, 480, 128, 160), (375, 1242, 96, 320), (120, 90, 64, 64), (33, 47, 64, 96)])
def test_warp_is_cv2_warpAffine_bit_for_bit(h, w, inp_h, inp_w):
    """Auto-skipping THIRD-PARTY pin of SURVEY rows a2 / f1: the reference's own call,
    `cv2.warpAffine(image, trans_input, (inp_width, inp_height), flags=cv2.INTER_LINEAR)` followed by
    `((inp / 255. - mean) / std).transpose(2, 0, 1)` (detector.py:218-224), against ct_preprocess_image on the crop of
    `_transform_scale` and on a rotated map.  cv2 is not installed in this image (the test skips); where it is, the
    fixed-point restatement of host, device and oracle is pinned to OpenCV itself."""
    cv2 = pytest.importorskip('cv2')
    rs = np.random.RandomState(h * 7 + w)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    meta = make_meta(inp_h, inp_w, h, w)
    rot = get_affine_transform(np.array([w / 2., h / 2.], np.float32), max(h, w) * 0.7, 25, [inp_w, inp_h])
    for t in (meta['trans_input'], rot):
        ref = cv2.warpAffine(img, np.asarray(t, np.float64), (inp_w, inp_h), flags=cv2.INTER_LINEAR)
        np.testing.assert_array_equal(_run(img, t, inp_w, inp_h), _norm(ref))
        np.testing.assert_array_equal(oimage.pre_process_image(img, t, inp_w, inp_h, MEAN, STD), _norm(ref))
