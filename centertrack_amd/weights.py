"""Synthetic (seeded) DLA-34 state dict with the reference's key schema.

There are no checkpoints offline, so tests and bench.py drive the model with
random-init weights of the exact architecture (SURVEY.md section 8d distributions): conv
weights ~ U(+-sqrt(3/fan_in)) (variance-preserving, so activations stay O(1)), BN statistics randomised so the
BN fold is exercised, DCN ``conv_offset_mask`` ~ N(0, off_std) so offsets are
non-zero, ``hm`` final bias = prior_bias (-4.6, reference opts.py:103).  Keys and
shapes follow the reference's ``DLASeg.state_dict()`` (SURVEY.md Appendix C;
reference src/lib/model/networks/dla.py, base_model.py) so real ``.pth`` files
load through the same path.
"""
import math
from collections import OrderedDict

import torch

CHANNELS = [16, 32, 64, 128, 256, 512]
LEVELS = [1, 1, 1, 2, 2, 1]

MOT_HEADS = OrderedDict([('hm', 1), ('reg', 2), ('wh', 2), ('tracking', 2), ('ltrb_amodal', 4)])
KITTI_HEADS = OrderedDict([('hm', 3), ('reg', 2), ('wh', 2), ('tracking', 2)])
COCO_HEADS = OrderedDict([('hm', 80), ('reg', 2), ('wh', 2), ('tracking', 2)])
# tracking,multi_pose (opts.py:343-354; COCO person key points: 17 joints)
POSE_HEADS = OrderedDict([('hm', 1), ('reg', 2), ('wh', 2), ('tracking', 2), ('hps', 34), ('hm_hp', 17),
                          ('hp_offset', 2)])
NUSC_HEADS = OrderedDict([('hm', 10), ('reg', 2), ('wh', 2), ('tracking', 2), ('dep', 1),
                          ('rot', 8), ('dim', 3), ('amodel_offset', 2)])


def dla34_param_shapes(heads, head_conv=256, pre_img=True, pre_hm=True):
    """Ordered list of (key, shape, kind) for every tensor of the reference DLASeg(34)
    state dict.  kind in {conv, bn, dcn_w, dcn_b, off_w, off_b, up, head_w, head_b,
    hm_out_b}."""
    out = []

    def conv(key, co, ci, k):
        out.append((key, (co, ci, k, k), 'conv'))

    def bn(prefix, c):
        out.append((prefix, (c,), 'bn'))

    def stem(prefix, ci):
        conv(prefix + '.0.weight', 16, ci, 7)
        bn(prefix + '.1', 16)

    def block(prefix, ci, co):
        conv(prefix + '.conv1.weight', co, ci, 3)
        bn(prefix + '.bn1', co)
        conv(prefix + '.conv2.weight', co, co, 3)
        bn(prefix + '.bn2', co)

    def tree(prefix, levels, ci, co, level_root, root_dim=0):
        # reference Tree.__init__, dla.py:175-213 (module registration order)
        if root_dim == 0:
            root_dim = 2 * co
        if level_root:
            root_dim += ci
        if levels == 1:
            block(prefix + '.tree1', ci, co)
            block(prefix + '.tree2', co, co)
            conv(prefix + '.root.conv.weight', co, root_dim, 1)
            bn(prefix + '.root.bn', co)
        else:
            tree(prefix + '.tree1', levels - 1, ci, co, False, 0)
            tree(prefix + '.tree2', levels - 1, co, co, False, root_dim + co)
        if ci != co:
            conv(prefix + '.project.0.weight', co, ci, 1)
            bn(prefix + '.project.1', co)

    stem('base.base_layer', 3)
    conv('base.level0.0.weight', 16, 16, 3)
    bn('base.level0.1', 16)
    conv('base.level1.0.weight', 32, 16, 3)
    bn('base.level1.1', 32)
    for i in range(2, 6):
        tree('base.level%d' % i, LEVELS[i], CHANNELS[i - 1], CHANNELS[i], i >= 3)
    if pre_img:
        stem('base.pre_img_layer', 3)
    if pre_hm:
        stem('base.pre_hm_layer', 1)

    def deform(prefix, ci, co):
        bn(prefix + '.actf.0', co)
        out.append((prefix + '.conv.weight', (co, ci, 3, 3), 'dcn_w'))
        out.append((prefix + '.conv.bias', (co,), 'dcn_b'))
        out.append((prefix + '.conv.conv_offset_mask.weight', (27, ci, 3, 3), 'off_w'))
        out.append((prefix + '.conv.conv_offset_mask.bias', (27,), 'off_b'))

    def ida(prefix, o, chans, ups):
        for i in range(1, len(chans)):
            deform('%s.proj_%d' % (prefix, i), chans[i], o)
            f = ups[i]
            out.append(('%s.up_%d.weight' % (prefix, i), (o, 1, 2 * f, 2 * f), 'up'))
            deform('%s.node_%d' % (prefix, i), o, o)

    ida('dla_up.ida_0', 256, [256, 512], [1, 2])
    ida('dla_up.ida_1', 128, [128, 256, 256], [1, 2, 2])
    ida('dla_up.ida_2', 64, [64, 128, 128, 128], [1, 2, 2, 2])
    ida('ida_up', 64, [64, 128, 256], [1, 2, 4])
    for h, c in heads.items():
        out.append((h + '.0.weight', (head_conv, 64, 3, 3), 'conv'))
        out.append((h + '.0.bias', (head_conv,), 'head_b'))
        out.append((h + '.2.weight', (c, head_conv, 1, 1), 'head_out_w:' + h))
        out.append((h + '.2.bias', (c,), 'hm_out_b' if 'hm' in h else 'head_b'))
    return out


def make_synthetic_state_dict(heads=None, seed=317, off_std=0.01, hm_gain=1.0,
                              prior_bias=-4.6, head_conv=256, dtype=torch.float32):
    """Seeded random DLA-34 weights (CPU tensors).  ``off_std`` = std of the DCN
    offset/mask conv weights (0.01: |offset| <~ 1 px; 0.1: stress, out-of-bounds taps).
    ``hm_gain`` scales the hm output layer so synthetic scores spread over (0,1) and the
    thresholded decode/tracker path is exercised."""
    heads = MOT_HEADS if heads is None else heads
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def uni(shape, bound):
        return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).mul_(bound).to(dtype)

    def nrm(shape, std):
        return torch.randn(shape, generator=g, dtype=torch.float64).mul_(std).to(dtype)

    for key, shape, kind in dla34_param_shapes(heads, head_conv):
        if kind == 'bn':
            sd[key + '.weight'] = (torch.rand(shape, generator=g, dtype=torch.float64) + 0.5).to(dtype)
            sd[key + '.bias'] = nrm(shape, 0.1)
            sd[key + '.running_mean'] = nrm(shape, 0.1)
            sd[key + '.running_var'] = (torch.rand(shape, generator=g, dtype=torch.float64) + 0.5).to(dtype)
            sd[key + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
        elif kind in ('conv', 'dcn_w'):
            # variance-preserving bounds (activations stay O(1) through ~50 layers so the
            # parity tests see real signal): U(+-sqrt(3/fan_in)); the DCN main conv gets
            # an extra x2 because the sigmoid mask (~0.5) halves its input.
            fan_in = shape[1] * shape[2] * shape[3]
            sd[key] = uni(shape, (2.0 if kind == 'dcn_w' else 1.0) * math.sqrt(3.0 / fan_in))
        elif kind.startswith('head_out_w'):
            gain = hm_gain if kind.endswith((':hm', ':hm_hp')) else 1.0
            sd[key] = uni(shape, gain / math.sqrt(shape[1]))
        elif kind in ('dcn_b', 'head_b'):
            sd[key] = nrm(shape, 0.1)
        elif kind == 'off_w':
            sd[key] = nrm(shape, off_std)
        elif kind == 'off_b':
            sd[key] = nrm(shape, 10 * off_std)
        elif kind == 'hm_out_b':
            sd[key] = torch.full(shape, prior_bias, dtype=dtype)
        elif kind == 'up':
            # bilinear kernel of reference fill_up_weights (dla.py:454-463), perturbed per
            # channel because the deconvs are *learned* in real checkpoints.
            k = shape[2]
            f = math.ceil(k / 2)
            c = (2 * f - 1 - f % 2) / (2. * f)
            base = torch.tensor([[(1 - abs(i / f - c)) * (1 - abs(j / f - c))
                                  for j in range(k)] for i in range(k)], dtype=torch.float64)
            w = base.view(1, 1, k, k) * (1 + 0.2 * torch.randn(shape, generator=g, dtype=torch.float64))
            sd[key] = w.to(dtype)
        else:
            raise ValueError(kind)
    return sd


def synthetic_inputs(batch, height, width, seed=317, n_blobs=8, dtype=torch.float32):
    """x, pre_img ~ N(0,1) [B,3,H,W] (normalised-image statistics) and a pre_hm with a few
    unit-peak Gaussians in [0,1]."""
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn((batch, 3, height, width), generator=g, dtype=torch.float64).to(dtype)
    pre = torch.randn((batch, 3, height, width), generator=g, dtype=torch.float64).to(dtype)
    hm = torch.zeros((batch, 1, height, width), dtype=torch.float64)
    ys = torch.arange(height, dtype=torch.float64).view(height, 1)
    xs = torch.arange(width, dtype=torch.float64).view(1, width)
    for b in range(batch):
        for _ in range(n_blobs):
            cy = float(torch.rand((), generator=g)) * (height - 1)
            cx = float(torch.rand((), generator=g)) * (width - 1)
            sig = 2.0 + 6.0 * float(torch.rand((), generator=g))
            blob = torch.exp(-((ys - round(cy)) ** 2 + (xs - round(cx)) ** 2) / (2 * sig * sig))
            hm[b, 0] = torch.maximum(hm[b, 0], blob)
    return x, pre, hm.to(dtype)
