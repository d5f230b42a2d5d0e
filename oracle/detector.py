"""Oracle: ``Detector.run`` for pre-processed inputs, on CPU.  TEST INFRASTRUCTURE ONLY.

Restates ``src/lib/detector.py``: run :55-172 (pre-processed-dict input path),
_transform_scale :175-204 + the meta part of pre_process :207-239, _trans_bbox
:242-251, _get_additional_inputs :254-290, _get_default_calib :293-297,
_sigmoid_output :300-308, _flip_output :311-332, process
:335-354 (the unconditional torch.cuda.synchronize() calls are dropped -- they
crash on a GPU-less host), post_process :356-369, merge_outputs :371-377,
reset_tracking :455-458.  Image warping (cv2.warpAffine) is out of the hot-path
scope: callers hand over the already normalised tensor, like test.py's
PrefetchDataset does.
"""
import copy
import math
import types

import numpy as np
import torch

from . import dla34
from .decode import generic_decode
from .image import (affine_transform, draw_umich_gaussian, gaussian_radius,
                    get_affine_transform)
from .post_process import generic_post_process
from .tracker import Tracker


def default_opt(**kw):
    """Fields of the reference's ``opt`` the hot path reads, with
    the tracking-task derivations of opts.py:280-289 applied."""
    o = types.SimpleNamespace(
        tracking=True, pre_img=True, pre_hm=True, zero_pre_hm=False, flip_test=False,
        K=100, track_thresh=0.3, out_thresh=-1.0, pre_thresh=-1.0, new_thresh=0.3,
        max_age=-1, hungarian=False, public_det=False, zero_tracking=False,
        depth_scale=1.0, down_ratio=4, fix_res=True, fix_short=0, pad=31,
        input_h=512, input_w=512, test_focal_length=-1, rest_focal_length=1200,
        num_classes=1)
    for k, v in kw.items():
        setattr(o, k, v)
    if o.tracking:                                  # opts.py:280-285
        o.out_thresh = max(o.track_thresh, o.out_thresh)
        o.pre_thresh = max(o.track_thresh, o.pre_thresh)
        o.new_thresh = max(o.track_thresh, o.new_thresh)
    return o


def transform_scale(opt, height, width, scale=1):
    """detector.py:175-204 without the cv2.resize"""
    new_height = int(height * scale)
    new_width = int(width * scale)
    if opt.fix_short > 0:
        if height < width:
            inp_height = opt.fix_short
            inp_width = (int(width / height * opt.fix_short) + 63) // 64 * 64
        else:
            inp_height = (int(height / width * opt.fix_short) + 63) // 64 * 64
            inp_width = opt.fix_short
        c = np.array([width / 2, height / 2], dtype=np.float32)
        s = np.array([width, height], dtype=np.float32)
    elif opt.fix_res:
        inp_height, inp_width = opt.input_h, opt.input_w
        c = np.array([new_width / 2., new_height / 2.], dtype=np.float32)
        s = max(height, width) * 1.0
    else:
        inp_height = (new_height | opt.pad) + 1
        inp_width = (new_width | opt.pad) + 1
        c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
        s = np.array([inp_width, inp_height], dtype=np.float32)
    return c, s, inp_width, inp_height


def make_meta(opt, height, width, calib=None):
    """The ``meta`` dict of detector.py:207-239 for an original image of height x width."""
    c, s, inp_width, inp_height = transform_scale(opt, height, width)
    out_height = inp_height // opt.down_ratio
    out_width = inp_width // opt.down_ratio
    focal = opt.rest_focal_length if opt.test_focal_length < 0 else opt.test_focal_length
    if calib is None:                               # detector.py:293-297
        calib = np.array([[focal, 0, width / 2, 0], [0, focal, height / 2, 0], [0, 0, 1, 0]])
    else:
        calib = np.array(calib, dtype=np.float32)
    return {'calib': calib, 'c': c, 's': s, 'height': height, 'width': width,
            'out_height': out_height, 'out_width': out_width,
            'inp_height': inp_height, 'inp_width': inp_width,
            'trans_input': get_affine_transform(c, s, 0, [inp_width, inp_height]),
            'trans_output': get_affine_transform(c, s, 0, [out_width, out_height])}


def trans_bbox(bbox, trans, width, height):
    """detector.py:242-251"""
    bbox = np.array(copy.deepcopy(bbox), dtype=np.float32)
    bbox[:2] = affine_transform(bbox[:2], trans)
    bbox[2:] = affine_transform(bbox[2:], trans)
    bbox[[0, 2]] = np.clip(bbox[[0, 2]], 0, width - 1)
    bbox[[1, 3]] = np.clip(bbox[[1, 3]], 0, height - 1)
    return bbox


def render_pre_hm(opt, dets, meta, with_hm=True):
    """detector.py:254-290 -> (input_hm [1|2,1,H,W] f32 tensor, output_inds [1,n] i64)"""
    trans_input, trans_output = meta['trans_input'], meta['trans_output']
    inp_width, inp_height = meta['inp_width'], meta['inp_height']
    out_width, out_height = meta['out_width'], meta['out_height']
    input_hm = np.zeros((1, inp_height, inp_width), dtype=np.float32)
    output_inds = []
    for det in dets:
        if det['score'] < opt.pre_thresh or det['active'] == 0:
            continue
        bbox = trans_bbox(det['bbox'], trans_input, inp_width, inp_height)
        bbox_out = trans_bbox(det['bbox'], trans_output, out_width, out_height)
        h, w = bbox[3] - bbox[1], bbox[2] - bbox[0]
        if h > 0 and w > 0:
            radius = gaussian_radius((math.ceil(h), math.ceil(w)))
            radius = max(0, int(radius))
            ct = np.array([(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2], dtype=np.float32)
            ct_int = ct.astype(np.int32)
            if with_hm:
                draw_umich_gaussian(input_hm[0], ct_int, radius)
            ct_out = np.array([(bbox_out[0] + bbox_out[2]) / 2,
                               (bbox_out[1] + bbox_out[3]) / 2], dtype=np.int32)
            output_inds.append(ct_out[1] * out_width + ct_out[0])
    if with_hm:
        input_hm = input_hm[np.newaxis]
        if opt.flip_test:
            input_hm = np.concatenate((input_hm, input_hm[:, :, :, ::-1]), axis=0)
        input_hm = torch.from_numpy(input_hm)
    output_inds = torch.from_numpy(np.array(output_inds, np.int64).reshape(1, -1))
    return input_hm, output_inds


def sigmoid_output(opt, output):
    """detector.py:300-308"""
    if 'hm' in output:
        output['hm'] = output['hm'].sigmoid_()
    if 'hm_hp' in output:
        output['hm_hp'] = output['hm_hp'].sigmoid_()
    if 'dep' in output:
        output['dep'] = 1. / (output['dep'].sigmoid() + 1e-6) - 1.
        output['dep'] *= opt.depth_scale
    return output


COCO_FLIP_IDX = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]   # coco_hp.py:18-19


def mirror_joints(x, flip_idx, offsets):
    """model/utils.py:33-50 (flip_lr / flip_lr_off): mirror a joint map left-right -- reverse the width axis,
    exchange the left/right joints and, for offset maps (``offsets``: channels are (dx, dy) per joint),
    negate dx."""
    out = torch.flip(x, [3]).clone()
    if offsets:
        out = out.view(out.shape[0], -1, 2, out.shape[2], out.shape[3])
        out[:, :, 0] *= -1
    src = out.clone()
    for a, b in flip_idx:
        out[:, a], out[:, b] = src[:, b], src[:, a]
    return out.reshape(x.shape)


def flip_output(output, flip_idx=COCO_FLIP_IDX):
    """detector.py:311-332"""
    average_flips = ['hm', 'wh', 'dep', 'dim']
    neg_average_flips = ['amodel_offset']
    single_flips = ['ltrb', 'nuscenes_att', 'velocity', 'ltrb_amodal', 'reg',
                    'hp_offset', 'rot', 'tracking', 'pre_hm']
    for head in output:
        if head in average_flips:
            output[head] = (output[head][0:1] + torch.flip(output[head][1:2], [3])) / 2
        if head in neg_average_flips:
            flipped = torch.flip(output[head][1:2], [3])
            flipped[:, 0::2] *= -1
            output[head] = (output[head][0:1] + flipped) / 2
        if head in single_flips:
            output[head] = output[head][0:1]
        if head == 'hps':
            output[head] = (output[head][0:1] + mirror_joints(output[head][1:2], flip_idx, True)) / 2
        if head == 'hm_hp':
            output[head] = (output[head][0:1] + mirror_joints(output[head][1:2], flip_idx, False)) / 2
    return output


class Detector(object):
    """CPU oracle of the reference Detector for pre-processed inputs."""

    def __init__(self, opt, state_dict, heads):
        self.opt = opt
        self.sd = state_dict
        self.heads = heads
        self.pre_images = None
        self.tracker = Tracker(opt.new_thresh, opt.max_age, opt.hungarian, opt.public_det)

    def process(self, images, pre_images=None, pre_hms=None, pre_inds=None):
        """detector.py:335-354"""
        with torch.no_grad():
            output = dla34.forward(images, pre_images, pre_hms, self.sd, self.heads)[-1]
            output = sigmoid_output(self.opt, output)
            output.update({'pre_inds': pre_inds})
            if self.opt.flip_test:
                output = flip_output(output)
            dets = generic_decode(output, K=self.opt.K, zero_tracking=self.opt.zero_tracking)
            for k in dets:
                dets[k] = dets[k].detach().cpu().numpy()
        return output, dets

    def post_process(self, dets, meta):
        """detector.py:356-369 (scale == 1)"""
        dets = generic_post_process(self.opt.out_thresh, dets, [meta['c']], [meta['s']],
                                    meta['out_height'], meta['out_width'], [meta['calib']])
        return dets[0]

    def merge_outputs(self, detections):
        """detector.py:371-377"""
        return [d for d in detections[0] if d['score'] > self.opt.out_thresh]

    def run(self, images, meta):
        """detector.py:55-172, pre-processed path, one scale; returns the result list
        (and keeps the raw decode in ``self.last_dets`` for parity tests)."""
        pre_hms, pre_inds = None, None
        if self.opt.tracking:
            if self.pre_images is None:
                self.pre_images = images
                self.tracker.init_track(meta['pre_dets'] if 'pre_dets' in meta else [])
            if self.opt.pre_hm:
                pre_hms, pre_inds = render_pre_hm(self.opt, self.tracker.tracks, meta,
                                                  with_hm=not self.opt.zero_pre_hm)
        output, dets = self.process(images, self.pre_images, pre_hms, pre_inds)
        self.last_output, self.last_dets, self.last_pre_hm = output, dets, pre_hms
        result = self.post_process(dets, meta)
        results = self.merge_outputs([result])
        if self.opt.tracking:
            public_det = meta['cur_dets'] if self.opt.public_det else None
            results = self.tracker.step(results, public_det)
            self.pre_images = images
        return results

    def reset_tracking(self):
        self.tracker.reset()
        self.pre_images = None
