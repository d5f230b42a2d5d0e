"""Oracle: DLA-34 + DLAUp + IDAUp + heads forward on CPU (functional torch fp32/fp64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
``src/lib/model/networks/dla.py`` (DLA :231-316, BasicBlock :38-66, Root :154-172,
Tree :174-228, DeformConv :506-518, IDAUp :520-545, DLAUp :549-574, DLASeg
:593-640) and ``base_model.py:73-91`` as pure functions over a state dict whose
keys are the reference's own, so reference checkpoints
drive it unchanged.  The DCN op inside DeformConv is oracle/dcn_v2.py (parity
unpinned for that op; everything else here is pinned against the imported
reference by tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F

from .dcn_v2 import dcn_forward

LEVELS = [1, 1, 1, 2, 2, 1]            # dla.py:333-335 (dla34)
CHANNELS = [16, 32, 64, 128, 256, 512]
BN_EPS = 1e-5                          # nn.BatchNorm2d default (dla.py:25,44)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.1, BN_EPS)


def _basic_block(x, sd, p, stride, residual=None):
    """dla.py:52-66"""
    if residual is None:
        residual = x
    out = F.conv2d(x, sd[p + '.conv1.weight'], None, stride=stride, padding=1)
    out = F.relu(_bn(out, sd, p + '.bn1'))
    out = F.conv2d(out, sd[p + '.conv2.weight'], None, stride=1, padding=1)
    out = _bn(out, sd, p + '.bn2')
    out = out + residual
    return F.relu(out)


def _root(xs, sd, p):
    """dla.py:164-172 (residual=False for dla34)"""
    x = F.conv2d(torch.cat(xs, 1), sd[p + '.conv.weight'], None)
    return F.relu(_bn(x, sd, p + '.bn'))


def _tree(x, sd, p, levels, cin, cout, stride, level_root, children=None):
    """dla.py:215-228.  Note the ``residual`` argument of the reference is
    always overwritten by ``project(bottom)``/``bottom`` (:218), so for
    levels > 1 the outer ``project`` result is dead code."""
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride=stride) if stride > 1 else x
    if cin != cout:
        residual = _bn(F.conv2d(bottom, sd[p + '.project.0.weight'], None), sd, p + '.project.1')
    else:
        residual = bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = _basic_block(x, sd, p + '.tree1', stride, residual)
        x2 = _basic_block(x1, sd, p + '.tree2', 1)
        return _root([x2, x1] + children, sd, p + '.root')
    x1 = _tree(x, sd, p + '.tree1', levels - 1, cin, cout, stride, False)
    children.append(x1)
    return _tree(x1, sd, p + '.tree2', levels - 1, cout, cout, 1, False, children)


def _stem(x, sd, p):
    return F.relu(_bn(F.conv2d(x, sd[p + '.0.weight'], None, padding=3), sd, p + '.1'))


def dla_base(x, pre_img, pre_hm, sd, p='base'):
    """DLA.forward, dla.py:305-316"""
    y = []
    x = _stem(x, sd, p + '.base_layer')
    if pre_img is not None:
        x = x + _stem(pre_img, sd, p + '.pre_img_layer')
    if pre_hm is not None:
        x = x + _stem(pre_hm, sd, p + '.pre_hm_layer')
    x = F.relu(_bn(F.conv2d(x, sd[p + '.level0.0.weight'], None, padding=1), sd, p + '.level0.1'))
    y.append(x)
    x = F.relu(_bn(F.conv2d(x, sd[p + '.level1.0.weight'], None, stride=2, padding=1), sd, p + '.level1.1'))
    y.append(x)
    for i in range(2, 6):
        x = _tree(x, sd, '%s.level%d' % (p, i), LEVELS[i], CHANNELS[i - 1], CHANNELS[i], 2, i >= 3)
        y.append(x)
    return y


def _deform_conv(x, sd, p):
    """DeformConv.forward, dla.py:515-518: DCN -> BN -> ReLU"""
    x = dcn_forward(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'],
                    sd[p + '.conv.conv_offset_mask.weight'], sd[p + '.conv.conv_offset_mask.bias'])
    return F.relu(_bn(x, sd, p + '.actf.0'))


def _ida_up(layers, startp, endp, sd, p, up_f):
    """IDAUp.forward, dla.py:539-545 (in-place on ``layers``).  ``up_f[i-startp]`` is
    the deconv factor of up_{i-startp}: kernel 2f, stride f, padding f//2, groups=o."""
    for i in range(startp + 1, endp):
        k = i - startp
        f = up_f[k]
        w = sd['%s.up_%d.weight' % (p, k)]
        x = _deform_conv(layers[i], sd, '%s.proj_%d' % (p, k))
        x = F.conv_transpose2d(x, w, None, stride=f, padding=f // 2, groups=w.shape[0])
        layers[i] = _deform_conv(x + layers[i - 1], sd, '%s.node_%d' % (p, k))


def dla_up(layers, sd, p='dla_up'):
    """DLAUp.forward, dla.py:568-574 with first_level=2 (channels 64..512, scales 1,2,4,8)."""
    layers = list(layers)
    out = [layers[-1]]
    ups = {0: [1, 2], 1: [1, 2, 2], 2: [1, 2, 2, 2]}   # dla.py:558-566
    for i in range(len(layers) - 2 - 1):
        _ida_up(layers, len(layers) - i - 2, len(layers), sd, '%s.ida_%d' % (p, i), ups[i])
        out.insert(0, layers[-1])
    return out


def dla_seg_features(x, pre_img, pre_hm, sd):
    """DLASeg.imgpre2feats / img2feats, dla.py:619-640"""
    y = dla_base(x, pre_img, pre_hm, sd)
    y = dla_up(y, sd)
    z = [y[0].clone(), y[1].clone(), y[2].clone()]
    _ida_up(z, 0, 3, sd, 'ida_up', [1, 2, 4])
    return z[-1]


def apply_heads(feat, heads, sd):
    """base_model.py:86-90 with head_conv=[256]: conv3x3+bias, ReLU, conv1x1+bias"""
    z = {}
    for head in heads:
        h = F.relu(F.conv2d(feat, sd[head + '.0.weight'], sd[head + '.0.bias'],
                            padding=sd[head + '.0.weight'].shape[-1] // 2))
        z[head] = F.conv2d(h, sd[head + '.2.weight'], sd[head + '.2.bias'])
    return z


def forward(x, pre_img, pre_hm, sd, heads):
    """BaseModel.forward, base_model.py:73-91 (num_stacks=1, model_output_list=False).
    Returns ``[ {head: tensor[B,c,H/4,W/4]} ]`` like the reference."""
    feat = dla_seg_features(x, pre_img, pre_hm, sd)
    return [apply_heads(feat, heads, sd)]
