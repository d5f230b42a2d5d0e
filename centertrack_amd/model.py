"""DLA-34 + DLAUp/IDAUp (DCNv2 nodes) + heads as a pre-planned sequence of HIP launches.

Mirrors the reference's model boundary: ``create_model(arch, heads,
head_conv, opt)`` / ``load_model`` (src/lib/model/model.py:24-90) and
``model(x, pre_img, pre_hm)[-1] -> {head: [B,c,h,w]}`` (base_model.py:73-91,
dla.py:593-640), with the reference's state-dict keys, so its ``.pth`` files load
unchanged.  Internally nothing of torch.nn runs: weights are BN-folded and packed
once into MFMA fragment order, activations are NHWC views, concatenations are
channel slices of shared buffers, and each frame is ~70 launches of
libcentertrack_hip kernels on the current stream (capturable in one HIP graph).
"""
import ctypes
import os
from collections import OrderedDict

import torch

from . import _lib, autotune, ops
from .ops import View
from .weights import CHANNELS, LEVELS, dla34_param_shapes

BN_EPS = 1e-5
SMALL_SLOT_WGS = 600      # a MAIN slot with fewer workgroups than this (of 768 resident ones) may split K finer
FUSE_OFFSET = os.environ.get('CENTERTRACK_FUSE_OFFSET', '1') != '0'
WINOGRAD = os.environ.get('CENTERTRACK_WINOGRAD', '1') != '0'
FUSE_HEADS = os.environ.get('CENTERTRACK_FUSE_HEADS', '1') != '0'
FOLD_POOL = os.environ.get('CENTERTRACK_FOLD_POOL', '1') != '0'       # 2x2 max-pools as side outputs of the stride-2 convs
FUSE_PROJ = os.environ.get('CENTERTRACK_FUSE_PROJ', '1') != '0'       # Tree.project computed by the tree1.conv1 launch (round 4)


def _fold_bn(sd, p):
    """eval-mode BatchNorm -> (scale, shift): y = x*scale + shift  """
    scale = sd[p + '.weight'].double() / torch.sqrt(sd[p + '.running_var'].double() + BN_EPS)
    shift = sd[p + '.bias'].double() - sd[p + '.running_mean'].double() * scale
    return scale.float().contiguous(), shift.float().contiguous()


class _Launch(object):
    """One pre-built C-ABI call of the frame (`keep` pins the tensors its raw pointers refer to)."""
    __slots__ = ('fn', 'args', 'name', 'keep', 'us', 'ws_need')

    def __init__(self, name, fn, args, keep=(), reads=(), writes=(), us=5.0, ws_need=0):
        self.name, self.fn, self.args, self.keep = name, fn, args, keep
        self.us, self.ws_need = us, ws_need


class _DcnLayer(object):
    """One DeformConv node of the IDAUp tree (dla.py:506-518) waiting to be scheduled: input view, output view, the
    fused IDAUp step ``up`` = (weight, f, skip view, output view) of a `proj` node."""
    __slots__ = ('name', 'x', 'cout', 'out', 'up', 'main', 'finish', 'desc', 'fused', 'splits', 'use_ws', 'nkk')

    def __init__(self, name, x, cout, out, up):
        self.name, self.x, self.cout, self.out, self.up = name, x, cout, out, up


class DLASegHIP(torch.nn.Module):
    """Drop-in for the reference ``DLASeg(34, heads, head_convs, opt)`` at inference."""

    def __init__(self, heads, head_conv=256, pre_img=True, pre_hm=True, depth_scale=1.0,
                 model_output_list=False):
        super().__init__()
        self.heads = OrderedDict(heads)
        if isinstance(head_conv, dict):       # reference passes {head: [256]}
            vals = {tuple(v) if isinstance(v, (list, tuple)) else (v,) for v in head_conv.values()}
            assert len(vals) == 1 and len(next(iter(vals))) == 1, 'only one head-conv layer is supported'
            head_conv = next(iter(vals))[0]
        self.head_conv = head_conv
        self.pre_img, self.pre_hm = pre_img, pre_hm
        self.depth_scale = depth_scale
        self.model_output_list = model_output_list
        # parameters/buffers registered under the reference's names so state_dict() matches
        self._names = []
        for key, shape, kind in dla34_param_shapes(self.heads, head_conv, pre_img, pre_hm):
            if kind == 'bn':
                for suf, val in (('weight', 1.0), ('bias', 0.0), ('running_mean', 0.0), ('running_var', 1.0)):
                    self._reg(key + '.' + suf, torch.full(shape, val))
                self._reg(key + '.num_batches_tracked', torch.tensor(0, dtype=torch.long))
            else:
                self._reg(key, torch.zeros(shape))
        self._prepared = None
        self._plans = {}

    def _reg(self, name, t):
        self.register_buffer(name.replace('.', '__'), t)
        self._names.append(name)

    # --- state dict with the reference's dotted keys ---------------------------------
    def state_dict(self, *a, **k):
        return OrderedDict((n, getattr(self, n.replace('.', '__'))) for n in self._names)

    def load_state_dict(self, sd, strict=True):
        missing = [n for n in self._names if n not in sd]
        unexpected = [k for k in sd if k not in set(self._names)]
        if strict and (missing or unexpected):
            raise RuntimeError('missing %s unexpected %s' % (missing[:5], unexpected[:5]))
        for n in self._names:
            if n in sd:
                buf = getattr(self, n.replace('.', '__'))
                if tuple(buf.shape) != tuple(sd[n].shape):
                    raise RuntimeError('shape mismatch for %s: %s vs %s' % (n, tuple(buf.shape), tuple(sd[n].shape)))
                buf.copy_(sd[n])
        self._prepared = None
        self._plans = {}
        return missing, unexpected

    # --- one-time weight preparation ----------------------------------------------------
    def _prepare(self):
        if self._prepared is not None:
            return self._prepared
        dev = next(self.buffers()).device
        if dev.type != 'cuda':
            raise _lib.CTError('DLASegHIP runs on an MI355X only: call .to("cuda") first (no CPU fallback)')
        _lib.load()
        sd = {k: v.to(dev) for k, v in self.state_dict().items()}
        P = {}

        def wino(w):
            """Winograd F(2x2,3x3) form of a 3x3 weight the stride-1 layers may be run with (autotuner's choice)"""
            ok = WINOGRAD and w.shape[2] == 3 and w.shape[1] % 64 == 0
            return ops.pack_winograd(w) if ok else None

        def conv_bn(wkey, bnkey):
            sc, sh = _fold_bn(sd, bnkey)
            return ops.pack_weight(sd[wkey]), sc, sh, wino(sd[wkey])

        P['stem_w'] = [sd['base.base_layer.0.weight'].contiguous(),
                       sd['base.pre_img_layer.0.weight'].contiguous() if self.pre_img else None,
                       sd['base.pre_hm_layer.0.weight'].contiguous() if self.pre_hm else None]
        sc3 = torch.ones(3, 16, device=dev)
        sh3 = torch.zeros(3, 16, device=dev)
        for i, (name, on) in enumerate((('base_layer', True), ('pre_img_layer', self.pre_img),
                                        ('pre_hm_layer', self.pre_hm))):
            if on:
                sc3[i], sh3[i] = _fold_bn(sd, 'base.%s.1' % name)
        P['stem_scale'], P['stem_shift'] = sc3.contiguous(), sh3.contiguous()
        P['level0'] = conv_bn('base.level0.0.weight', 'base.level0.1')
        P['level1'] = conv_bn('base.level1.0.weight', 'base.level1.1')

        def leaf(p, cin, cout):
            d = {'c11': conv_bn(p + '.tree1.conv1.weight', p + '.tree1.bn1'),
                 'c12': conv_bn(p + '.tree1.conv2.weight', p + '.tree1.bn2'),
                 'c21': conv_bn(p + '.tree2.conv1.weight', p + '.tree2.bn1'),
                 'c22': conv_bn(p + '.tree2.conv2.weight', p + '.tree2.bn2'),
                 'root': conv_bn(p + '.root.conv.weight', p + '.root.bn')}
            if cin != cout:
                d['proj'] = conv_bn(p + '.project.0.weight', p + '.project.1')
            return d

        for i in range(2, 6):
            p = 'base.level%d' % i
            if LEVELS[i] == 1:
                P[p] = leaf(p, CHANNELS[i - 1], CHANNELS[i])
            else:
                P[p + '.tree1'] = leaf(p + '.tree1', CHANNELS[i - 1], CHANNELS[i])
                P[p + '.tree2'] = leaf(p + '.tree2', CHANNELS[i], CHANNELS[i])

        def deform(p):
            sc, sh = _fold_bn(sd, p + '.actf.0')
            sh = (sd[p + '.conv.bias'].double() * sc.double() + sh.double()).float().contiguous()
            return {'w': ops.pack_weight(sd[p + '.conv.weight']), 'scale': sc, 'shift': sh,
                    'w_off': ops.pack_weight(sd[p + '.conv.conv_offset_mask.weight']),
                    'w_off_wino': wino(sd[p + '.conv.conv_offset_mask.weight']),
                    'b_off': sd[p + '.conv.conv_offset_mask.bias'].contiguous()}

        for p, n in (('dla_up.ida_0', 1), ('dla_up.ida_1', 2), ('dla_up.ida_2', 3), ('ida_up', 2)):
            for k in range(1, n + 1):
                P['%s.proj_%d' % (p, k)] = deform('%s.proj_%d' % (p, k))
                P['%s.node_%d' % (p, k)] = deform('%s.node_%d' % (p, k))
                P['%s.up_%d' % (p, k)] = ops.upsample_weight(sd['%s.up_%d.weight' % (p, k)])
        # heads with <= 8 output channels: ONE launch for conv3x3 + ReLU + conv1x1 of all of them (ct_heads_fused)
        hc = self.head_conv
        # ('hm_hp' stays out whatever its width: ct_heads_fused has ONE sigmoid range, taken by 'hm'; the per-head 1x1
        # tail applies hm_hp's sigmoid, detector.py:300-304)
        small = ([h for h, c in self.heads.items() if c <= 8 and h != 'hm_hp']
                 if (FUSE_HEADS and WINOGRAD and hc == 256) else [])
        small = small[:_lib.CT_MAX_FUSED_HEADS]
        P['heads_small'] = small
        def fused(names):
            """(w0 Winograd-packed, b0, w2 [n,8,256], b2 [n,8]) of ct_heads_fused for the heads ``names``"""
            w0s = torch.cat([sd[h + '.0.weight'] for h in names], 0)
            w2s = torch.zeros((len(names), 8, hc), device=dev)
            b2s = torch.zeros((len(names), 8), device=dev)
            for i, h in enumerate(names):
                c = self.heads[h]
                w2s[i, :c] = sd[h + '.2.weight'].reshape(c, hc)
                b2s[i, :c] = sd[h + '.2.bias']
            return (ops.pack_winograd(w0s), torch.cat([sd[h + '.0.bias'] for h in names], 0).contiguous(),
                    w2s.contiguous(), b2s.contiguous())

        if small:
            P['hs:' + ','.join(small)] = fused(small)
            if 'hm' in small and small != ['hm']:
                P['hs:hm'] = fused(['hm'])             # (sparse-heads plans: the dense launch holds 'hm' alone)
        big = [h for h in self.heads if h not in small]
        P['heads_big'] = big
        if small and big:                  # the remaining (wide) heads keep the two-launch form on their own channels
            w0b = torch.cat([sd[h + '.0.weight'] for h in big], 0)
            P['hb_w'], P['hb_ww'] = ops.pack_weight(w0b), wino(w0b)
            P['hb_b'] = torch.cat([sd[h + '.0.bias'] for h in big], 0).contiguous()
        # heads: all first layers share their input -> one 64 -> 256*nh conv
        w0 = torch.cat([sd[h + '.0.weight'] for h in self.heads], 0)
        P['head0_w'] = ops.pack_weight(w0)
        P['head0_ww'] = wino(w0)
        P['head0_b'] = torch.cat([sd[h + '.0.bias'] for h in self.heads], 0).contiguous()
        # ... and all second (1x1) layers as ONE block-diagonal 256*nh -> sum(c) conv: a single launch reads
        # the intermediate once and writes every head into channel slices of one NCHW tensor
        hc = self.head_conv
        ctot = sum(self.heads.values())
        w2 = torch.zeros((ctot, hc * len(self.heads), 1, 1), device=dev)
        c0 = 0
        for j, (h, c) in enumerate(self.heads.items()):
            w2[c0:c0 + c, hc * j:hc * (j + 1)] = sd[h + '.2.weight']
            c0 += c
        P['head2_w'] = ops.pack_weight(w2)
        P['head2_b'] = torch.cat([sd[h + '.2.bias'] for h in self.heads], 0).contiguous()
        for h in self.heads:      # per-head form, used when the block-diagonal one would waste too many flops
            P[h + '.2'] = (ops.pack_weight(sd[h + '.2.weight']), sd[h + '.2.bias'].contiguous())
        # sparse heads (round 5, opt-in): the regression heads the decode only reads at the K winners -- direct-form
        # packed conv3x3 weights (the Winograd form has no single-pixel evaluation), hidden bias, [c, 256] output layer
        if hc == 256:
            for h, c in self.heads.items():
                if h in _lib.HEAD_INDEX:
                    P['sparse.' + h] = (ops.pack_weight(sd[h + '.0.weight']), sd[h + '.0.bias'].contiguous(),
                                        sd[h + '.2.weight'].reshape(c, hc).contiguous(), sd[h + '.2.bias'].contiguous())
        self._prepared = P
        return P

    def _build_plan(self, N, H, W, with_img, with_hm, fuse_sigmoid, sparse_heads=False):
        P = self._prepare()
        lib = _lib.load()
        dev = next(self.buffers()).device
        if H % 32 or W % 32:
            raise _lib.CTError('input %dx%d must be a multiple of 32 (reference pads to 32 too, opts.py:297)' % (H, W))
        plan = {'launches': [], 'ws_need': 0}
        L = plan['launches']
        tune = autotune.enabled()

        def alloc(h, w, c):
            return ops.new_view(N, h, w, c, dev)

        def add_conv(name, x, pk, cout, ks, stride=1, relu=True, res=None, out=None, **kw):
            wp, sc, sh, ww = pk
            d = ops.make_conv_desc(x, wp, cout, ks, stride, scale=sc, shift=sh, res=res, relu=relu, out=out,
                                   w_wino=(ww if stride == 1 else None), **kw)
            us = autotune.tune_conv(d, dev)[2] if tune else 10.0
            L.append(_Launch(name, 'conv', d, (x, res, out, pk, kw.get('out_nchw'), kw.get('pool'), kw.get('proj')), reads=(x, res),
                             writes=(out, kw.get('out_nchw')), us=us,
                             ws_need=lib.ct_conv2d_workspace_bytes(ctypes.byref(d))))
            return out

        # static inputs (copied into before each replay)
        x_in = torch.zeros((N, 3, H, W), device=dev)
        img_in = torch.zeros((N, 3, H, W), device=dev) if with_img else None
        hm_in = torch.zeros((N, 1, H, W), device=dev) if with_hm else None
        plan['inputs'] = (x_in, img_in, hm_in)
        s0 = alloc(H, W, 16)
        L.append(_Launch('stem', 'stem', (x_in, img_in, hm_in, s0), (), reads=(x_in, img_in, hm_in), writes=(s0,), us=46.0))
        l0 = add_conv('level0', s0, P['level0'], 16, 3, out=alloc(H, W, 16))
        l1 = add_conv('level1', l0, P['level1'], 32, 3, stride=2, out=alloc(H // 2, W // 2, 32))

        def leaf(name, x, pk, cin, cout, stride, R, out, bottom=None, level_root=False, make_bottom=False):
            """Tree(levels=1).forward (dla.py:215-228) over the concat buffer R = [x2 | x1 | children].  ``bottom`` =
            Tree.downsample(x) (dla.py:207,217): a 2x2 max-pool that (round 3) tree1.conv1 -- the 3x3 stride-2 conv over
            the same x -- writes as a side output of its launch instead of a launch of its own (``make_bottom``: the
            caller owns the buffer and wants it filled here)."""
            h, w = x.H // stride, x.W // stride
            fold = None
            # round 4: Tree.project (conv1x1 + BN of the pooled input, dla.py:196-203,217-218) is a second output of the
            # tree1.conv1 launch -- same input patches, the pooling window is four taps of the 3x3 stride-2 window
            fuse_proj = FUSE_PROJ and stride == 2 and cin != cout
            if stride > 1 and bottom is None:
                if level_root:
                    bottom = R.slice(2 * cout, cin)
                    make_bottom = True
                elif not fuse_proj:
                    bottom = alloc(h, w, cin)
                    make_bottom = True
                # (else: the pooled tensor only feeds the projection and is never materialised)
            elif stride == 1:
                bottom = x
            if stride > 1 and make_bottom:
                if FOLD_POOL:
                    fold = bottom
                else:
                    L.append(_Launch(name + '.pool', 'pool', (x, bottom), reads=(x,), writes=(bottom,)))
            t = alloc(h, w, cout)
            x1 = R.slice(cout, cout)
            x2 = R.slice(0, cout)
            if fuse_proj:
                residual = alloc(h, w, cout)
                add_conv(name + '.t1.conv1+project', x, pk['c11'], cout, 3, stride=stride, out=t, pool=fold,
                         proj=(pk['proj'][0], pk['proj'][1], pk['proj'][2], residual))
            else:
                add_conv(name + '.t1.conv1', x, pk['c11'], cout, 3, stride=stride, out=t, pool=fold)
                if cin != cout:
                    residual = add_conv(name + '.project', bottom, pk['proj'], cout, 1, relu=False, out=alloc(h, w, cout))
                else:
                    residual = bottom
            add_conv(name + '.t1.conv2', t, pk['c12'], cout, 3, res=residual, out=x1)
            add_conv(name + '.t2.conv1', x1, pk['c21'], cout, 3, out=t)
            add_conv(name + '.t2.conv2', t, pk['c22'], cout, 3, res=x1, out=x2)
            add_conv(name + '.root', View(R.buf, R.c0, R.C), pk['root'], cout, 1, out=out)
            return out

        feats = [l0, l1]
        x = l1
        for i in range(2, 6):
            p = 'base.level%d' % i
            cin, cout = CHANNELS[i - 1], CHANNELS[i]
            h, w = x.H // 2, x.W // 2
            out = alloc(h, w, cout)
            level_root = i >= 3
            if LEVELS[i] == 1:
                R = alloc(h, w, 2 * cout + (cin if level_root else 0))
                leaf(p, x, P[p], cin, cout, 2, R, out, level_root=level_root)
            else:
                # Tree(levels=2): R2 = [y2 | y1 | bottom | x1]; the outer project is dead code (dla.py:218)
                R2 = alloc(h, w, 3 * cout + cin)
                bottom = R2.slice(2 * cout, cin)
                x1 = R2.slice(2 * cout + cin, cout)
                R1 = alloc(h, w, 2 * cout)
                leaf(p + '.tree1', x, P[p + '.tree1'], cin, cout, 2, R1, x1, bottom=bottom, make_bottom=True)
                leaf(p + '.tree2', x1, P[p + '.tree2'], cout, cout, 1, R2, out)
            feats.append(out)
            x = out

        dcn_layers = []

        def deform(name, x, cout, out, up=None):
            """DeformConv.forward (dla.py:515-518): offset/mask conv -> DCNv2 -> BN -> ReLU; with ``up`` =
            (weight, f, skip view, output view) also the IDAUp step `up(.) + skip` (dla.py:543-545) that consumes a
            `proj` node.  Only recorded here: the 16 nodes are scheduled together below (_schedule_dcn)."""
            dcn_layers.append(_DcnLayer(name, x, cout, out, up))
            return out

        def ida(p, layers, startp, endp, o, up_f):
            """IDAUp.forward (dla.py:539-545)"""
            for i in range(startp + 1, endp):
                k = i - startp
                f = up_f[k]
                xi = layers[i]
                up = alloc(xi.H * f, xi.W * f, o)
                deform('%s.proj_%d' % (p, k), xi, o, alloc(xi.H, xi.W, o),
                       up=(P['%s.up_%d' % (p, k)], f, layers[i - 1], up))
                layers[i] = deform('%s.node_%d' % (p, k), up, o, alloc(up.H, up.W, o))

        layers = list(feats)                                   # DLAUp.forward, dla.py:568-574
        outs = [layers[-1]]
        ida('dla_up.ida_0', layers, 4, 6, 256, [1, 2]); outs.insert(0, layers[-1])
        ida('dla_up.ida_1', layers, 3, 6, 128, [1, 2, 2]); outs.insert(0, layers[-1])
        ida('dla_up.ida_2', layers, 2, 6, 64, [1, 2, 2, 2]); outs.insert(0, layers[-1])
        y = [outs[0], outs[1], outs[2]]                        # DLASeg.imgpre2feats, dla.py:631-640
        ida('ida_up', y, 0, 3, 64, [1, 2, 4])
        feat = y[-1]
        plan['feat'] = feat
        produced0 = {id(f_.buf): 0 for f_ in feats}
        knobs, dcn_launches = self._tune_dcn_schedule(dcn_layers, produced0, N, H, W, dev, tune)
        plan['dcn_knobs'] = knobs
        plan['dcn_layers'] = {ly.name: (ly.up[3] if ly.up is not None else ly.out) for ly in dcn_layers}   # name -> result view
        L.extend(dcn_launches)

        small, big = P['heads_small'], P['heads_big']
        plan['sparse'] = None
        if sparse_heads:
            # opt-in (detector only, never Module.forward): the regression heads the decode reads at the K winners are not
            # computed as maps -- ct_decode evaluates them at those pixels (ct_sparse_heads_desc)
            sp = [h for h in self.heads if ('sparse.' + h) in P]
            if {'ltrb', 'ltrb_amodal'} & set(self.heads):
                # dead heads: generic_decode overwrites the wh box with the ltrb / ltrb_amodal box (decode.py:131-139,150-159)
                # and `reg` only feeds the wh box (decode.py:102-110) -- neither reaches the returned dict, so with the
                # MOT head set (opts.py:343-359 + --ltrb_amodal) they are not evaluated at all
                sp = [h for h in sp if h not in ('reg', 'wh')]
            if not sp or 'hm' not in self.heads or {'hps', 'hm_hp'} & set(self.heads) or not fuse_sigmoid:
                raise _lib.CTError('sparse heads need an hm head, regression heads with head_conv 256, no pose heads '
                                   'and the detector\'s fused epilogues')
            plan['sparse'] = {'feat': feat, 'heads': [(h,) + P['sparse.' + h] for h in sp], 'depth_scale': self.depth_scale}
            small = [h for h in small if ('sparse.' + h) not in P]          # (the dead heads leave the dense launch too)
            if any(('sparse.' + h) in P for h in big):          # (more small heads than one fused launch holds: not with sparse heads)
                raise _lib.CTError('sparse heads: %s do not fit the fused-heads launch' % [h for h in big if ('sparse.' + h) in P])
        if small or plan['sparse'] is not None:
            outputs = self._plan_heads_fused(plan, L, P, feat, N, dev, tune, fuse_sigmoid, small, big)
        else:
            nh = len(self.heads)
            hc = self.head_conv
            mid = alloc(feat.H, feat.W, hc * nh)
            d = ops.make_conv_desc(feat, P['head0_w'], hc * nh, 3, 1, shift=P['head0_b'], relu=True, out=mid,
                                   w_wino=P['head0_ww'])
            us = autotune.tune_conv(d, dev)[2] if tune else 200.0
            L.append(_Launch('heads.0', 'conv', d, (feat, mid), reads=(feat,), writes=(mid,), us=us,
                             ws_need=lib.ct_conv2d_workspace_bytes(ctypes.byref(d))))
            ctot = sum(self.heads.values())
            comb = torch.empty((N, ctot, feat.H, feat.W), device=dev)
            sig, dep = (0, 0), (0, 0)
            c0 = 0
            outputs = OrderedDict()
            for hname, c in self.heads.items():
                if fuse_sigmoid and hname == 'hm':                 # (hm_hp -> per-head tail below: one sigmoid range per launch)
                    sig = (c0, c0 + c)
                if fuse_sigmoid and hname == 'dep':
                    dep = (c0, c0 + c)
                outputs[hname] = comb[:, c0:c0 + c]            # channel-slice views of the combined tensor
                c0 += c
            if ctot <= 32 and 'hm_hp' not in self.heads:
                # few output channels (MOT 11, KITTI 9, nuScenes 30): one block-diagonal conv over the whole intermediate
                d = ops.make_conv_desc(mid, P['head2_w'], ctot, 1, 1, shift=P['head2_b'], out_nchw=comb, sig=sig, dep=dep,
                                       depth_scale=self.depth_scale)
                us = autotune.tune_conv(d, dev)[2] if tune else 20.0
                L.append(_Launch('heads.2', 'conv', d, (mid, comb), reads=(mid,), writes=(comb,), us=us,
                                 ws_need=lib.ct_conv2d_workspace_bytes(ctypes.byref(d))))
            else:
                # a wide head (COCO: 80 classes): the block-diagonal form would multiply its flops by the number of heads
                outputs = OrderedDict()
                for j, (hname, c) in enumerate(self.heads.items()):
                    o = torch.empty((N, c, feat.H, feat.W), device=dev)
                    wp, b = P[hname + '.2']
                    hsig = (0, c) if (fuse_sigmoid and hname in ('hm', 'hm_hp')) else (0, 0)   # detector.py:300-304
                    hdep = (0, c) if (fuse_sigmoid and hname == 'dep') else (0, 0)
                    d = ops.make_conv_desc(mid.slice(hc * j, hc), wp, c, 1, 1, shift=b, out_nchw=o, sig=hsig, dep=hdep,
                                           depth_scale=self.depth_scale)
                    us = autotune.tune_conv(d, dev)[2] if tune else 7.0
                    L.append(_Launch('heads.%s.2' % hname, 'conv', d, (mid, o), reads=(mid.slice(hc * j, hc),), writes=(o,),
                                     us=us, ws_need=lib.ct_conv2d_workspace_bytes(ctypes.byref(d))))
                    outputs[hname] = o
        plan['outputs'] = outputs
        # split-K partials of the dense convs: one workspace shared by all of them (the launches of a frame are
        # sequential); every DCN layer owns its workspace (the layers of a group run concurrently)
        need = max([l.ws_need for l in L if l.fn == 'conv'] + [16])
        ws = torch.empty(need // 4, dtype=torch.float32, device=dev)
        plan['ws'] = [ws]
        for l in L:
            if l.fn == 'conv':
                l.args.workspace, l.args.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        plan['ws_need'] = ws.numel() * 4
        return plan

    def _plan_heads_fused(self, plan, L, P, feat, N, dev, tune, fuse_sigmoid, small, big):
        """Heads (base_model.py:24-65,86-90): every head with <= 8 output channels in ONE ct_heads_fused launch (hidden
        256-channel maps stay in the workgroups); wider heads (80-class hm, hps, hm_hp) as conv3x3 + per-head conv1x1."""
        lib = _lib.load()
        hc = self.head_conv
        outputs = {}
        if small:
            ctot = sum(self.heads[h] for h in small)
            comb = torch.empty((N, ctot, feat.H, feat.W), device=dev)
            hd = _lib.HeadsDesc()
            hd.x, hd.N, hd.H, hd.W, hd.Cin, hd.ldx = feat.ptr, N, feat.H, feat.W, feat.C, feat.ld
            hs_w0, hs_b0, hs_w2, hs_b2 = P['hs:' + ','.join(small)]
            hd.w0_winograd, hd.b0, hd.nheads = hs_w0.data_ptr(), hs_b0.data_ptr(), len(small)
            hd.w2, hd.b2, hd.out, hd.ctot = hs_w2.data_ptr(), hs_b2.data_ptr(), comb.data_ptr(), ctot
            hd.depth_scale = self.depth_scale
            c0 = 0
            for i, h in enumerate(small):
                c = self.heads[h]
                hd.cout[i], hd.coff[i] = c, c0
                if fuse_sigmoid and h == 'hm':
                    hd.sig_lo, hd.sig_hi = c0, c0 + c
                if fuse_sigmoid and h == 'dep':
                    hd.dep_lo, hd.dep_hi = c0, c0 + c
                outputs[h] = comb[:, c0:c0 + c]
                c0 += c
            L.append(_Launch('heads.fused[%s]' % ' + '.join(small), 'heads', hd, (feat, comb, hs_w0, hs_b0, hs_w2, hs_b2),
                             us=100.0))
        if big:
            mid = ops.new_view(N, feat.H, feat.W, hc * len(big), dev)
            d = ops.make_conv_desc(feat, P['hb_w'], hc * len(big), 3, 1, shift=P['hb_b'], relu=True, out=mid, w_wino=P['hb_ww'])
            us = autotune.tune_conv(d, dev)[2] if tune else 100.0
            L.append(_Launch('heads.0', 'conv', d, (feat, mid), us=us, ws_need=lib.ct_conv2d_workspace_bytes(ctypes.byref(d))))
            for j, h in enumerate(big):
                c = self.heads[h]
                o = torch.empty((N, c, feat.H, feat.W), device=dev)
                wp, b = P[h + '.2']
                hsig = (0, c) if (fuse_sigmoid and h in ('hm', 'hm_hp')) else (0, 0)       # detector.py:300-304
                hdep = (0, c) if (fuse_sigmoid and h == 'dep') else (0, 0)
                d = ops.make_conv_desc(mid.slice(hc * j, hc), wp, c, 1, 1, shift=b, out_nchw=o, sig=hsig, dep=hdep,
                                       depth_scale=self.depth_scale)
                us = autotune.tune_conv(d, dev)[2] if tune else 7.0
                L.append(_Launch('heads.%s.2' % h, 'conv', d, (mid, o), us=us,
                                 ws_need=lib.ct_conv2d_workspace_bytes(ctypes.byref(d))))
                outputs[h] = o
        return OrderedDict((h, outputs[h]) for h in self.heads if h in outputs)

    def run_stem_partial(self, x, pre_img, out):
        """the terms of the stem that do not depend on the tracker (dla.py:305-311: base_layer(x) + pre_img_layer(
        pre_img)) into the NHWC view ``out`` -- a detector runs this for frame t+1 while the host still associates
        frame t; ``_run_plan(..., stem_partial=out)`` then only adds the pre_hm term (bit-identical to one launch)"""
        P = self._prepare()
        lib = _lib.load()
        w = P['stem_w']
        rc = lib.ct_stem_forward_parts(x.data_ptr(), ops._p(pre_img), None, None, 0, x.shape[0], x.shape[2], x.shape[3],
                                       w[0].data_ptr(), ops._p(w[1]) if pre_img is not None else None, None,
                                       P['stem_scale'].data_ptr(), P['stem_shift'].data_ptr(), out.ptr, out.ld,
                                       _lib.stream_ptr())
        if rc != 0:
            _lib.check(rc, 'stem (x / pre_img terms)')

    def _run_plan(self, plan, inputs=None, stem_partial=None, launches=None):
        """Enqueue every launch of the plan on the current stream.  ``inputs`` = (x, pre_img, pre_hm) tensors to read
        instead of the plan's own static input buffers (a detector rotates its frame buffers so that the
        previous frame never has to be copied).  ``stem_partial``: NHWC view that already holds the x / pre_img terms
        of the stem (``run_stem_partial``): only the pre_hm term is computed, on top of it."""
        P = self._prepared
        lib = _lib.load()
        st = _lib.stream_ptr()
        for l in (plan['launches'] if launches is None else launches):      # (``launches``: a contiguous part of the plan, bench.py)
            if l.fn == 'conv':
                rc = lib.ct_conv2d(ctypes.byref(l.args), st)
            elif l.fn == 'dcn_group':
                arr, n, phases = l.args
                rc = lib.ct_dcn_v2_group(arr, n, phases, st)
            elif l.fn == 'heads':
                rc = lib.ct_heads_fused(ctypes.byref(l.args), st)
            elif l.fn == 'pool':
                x, y = l.args
                rc = lib.ct_maxpool2x2(x.ptr, x.N, x.H, x.W, x.C, x.ld, y.ptr, y.ld, st)
            elif l.fn == 'stem':
                x, img, hm, y = l.args
                if inputs is not None:
                    x, img, hm = inputs
                w = P['stem_w']
                if stem_partial is not None:
                    if hm is None:
                        raise _lib.CTError('a split stem needs the pre_hm input')
                    rc = lib.ct_stem_forward_parts(None, None, hm.data_ptr(), stem_partial.ptr, stem_partial.ld,
                                                   x.shape[0], x.shape[2], x.shape[3], None, None, w[2].data_ptr(),
                                                   P['stem_scale'].data_ptr(), P['stem_shift'].data_ptr(), y.ptr, y.ld, st)
                    if rc != 0:
                        _lib.check(rc, l.name)
                    continue
                rc = lib.ct_stem_forward(x.data_ptr(), ops._p(img), ops._p(hm), x.shape[0], x.shape[2], x.shape[3],
                                         w[0].data_ptr(), ops._p(w[1]) if img is not None else None,
                                         ops._p(w[2]) if hm is not None else None,
                                         P['stem_scale'].data_ptr(), P['stem_shift'].data_ptr(), y.ptr, y.ld, st)
            else:
                raise AssertionError(l.fn)
            if rc != 0:
                _lib.check(rc, l.name)

    # --- the 16 DeformConv nodes of DLAUp / IDAUp as grouped launches -------------------------------------
    def _schedule_dcn(self, layers, produced0, N, dev, knobs, tune=True):
        """Launch list of the 16 DCN nodes for one choice of ``knobs``.

        Dataflow (dla.py:539-574): a `proj` node only reads a finished level; the IDAUp step behind it needs the
        previous node's output as its skip tensor; a `node` reads that step's result.  Every layer is split in two
        launches' worth of work -- MAIN (gather + contraction, results or split-K partials) and FINISH (reduction +
        BN + ReLU + IDAUp step) -- and placed in the earliest slot t (M_t at time 2t, F_t at time 2t+1) whose
        inputs exist: main(L) = first t with 2t > time(x); finish(L) = first t' >= main(L) with 2t'+1 > time(skip).
        The layers sharing a slot go out as ONE ct_dcn_v2_group launch each for M_t and F_t: 8 + 7 launches for the
        16 nodes instead of 16 + (offset convs) + 8 reductions, and at one stream 768 .. 1536 workgroups per launch
        instead of 128 .. 512.  knobs = (fuse_max_cin, chunks_per_split, nkk, split_offsets): the offset/mask conv
        is computed inside the DCN launch for Cin <= fuse_max_cin -- every workgroup of a split-K layer then repeats
        it over ALL input channels, 47 % of a 256-channel layer's time at one stream -- else ahead of the slot: K-split
        over 64-channel chunks in ONE CT_DCN_OFFSETS launch for all layers of the slot (split_offsets = 1, one
        workgroup per 32-pixel tile and chunk whatever Cin; split_offsets = 3, round 6: the same launch as Winograd
        F(2x2,3x3) tiles, one workgroup per 64-pixel block and chunk) or by one conv launch per layer (0; 2: a Winograd
        launch per layer writing raw sums); every workgroup
        contracts chunks_per_split 32-channel chunks, in steps of 16 * nkk channels (nkk = 2 / 4: 16 / 32 MFMAs per
        wave between barriers).  knobs[4] (round 3) = chunks per split of the SMALL slots: at one stream the three `proj`
        slots hold 256 .. 384 workgroups of 18 (chunk, tap) steps each on 768 workgroup slots -- one wave per SIMD,
        nothing to hide the gather latency behind -- while their layers go through the workspace anyway (IDAUp step in
        the finishing launch); splitting K twice as fine fills the chip and halves every workgroup's loop.  0 = off,
        2 = two chunks per split, 1 = one chunk per split (then in 32-channel steps).  knobs[5] = 1: the layers of
        those slots also leave their offset convs to the slot's K-split OFFSETS launch (a fused conv would be repeated
        by every split)."""
        lib = _lib.load()
        P = self._prepared
        fuse_max_cin, cps, nkk = knobs[:3]
        split_offsets = knobs[3] if len(knobs) > 3 else 1
        cps_small = knobs[4] if len(knobs) > 4 else 0
        small_unfuse = knobs[5] if len(knobs) > 5 else 0      # small slots: offset convs out of the MAIN launch too
        # knobs[6] (round 6) = 1: MAIN launches as persistent launches (ct_dcn_desc.algo 5xxxx / 6xxxx: `dcn_slots` resident
        # workgroups striding over the pixel tiles of their column, gathers / weights / the next sampling table running
        # across tile boundaries) wherever every layer of the launch can -- no fused offset conv, an even number of step
        # units per split; bit-identical to the one-workgroup-per-tile launch
        persist = knobs[6] if len(knobs) > 6 else 0
        slot_cps = {}
        sizes, mains = self._dcn_slot_sizes(layers, produced0, N, cps, with_mains=True)
        if cps_small and cps_small < cps:
            for t, wgs in sizes.items():
                if wgs < SMALL_SLOT_WGS:
                    slot_cps[t] = cps_small
        # knobs[5] == 2: every slot that needs an OFFSETS launch anyway (a layer above the fuse threshold) leaves ALL its
        # offset convs to that launch -- a fused conv is repeated by every cout block and split of its layer
        unfuse_slots = set()
        if small_unfuse == 2:
            unfuse_slots = {m for ly, m in zip(layers, mains) if not (ly.x.C % 64 == 0 and ly.x.C <= fuse_max_cin and FUSE_OFFSET)}
        time_of = dict(produced0)                    # buffer id -> time after which it is readable

        def t_of(view):
            return time_of.get(id(view.buf), 0)

        slots = {}
        convs = {}
        offs = {}
        for ly in layers:                            # (reference order: producers come first)
            pk = P[ly.name]
            nchunks = ly.x.C // 32
            ly.main = t_of(ly.x) // 2 + 1
            c = slot_cps.get(ly.main, cps)
            # (a finely split layer would repeat a fused offset conv in every split: knobs[5] moves it to the slot's
            #  K-split OFFSETS launch instead)
            ly.fused = (ly.x.C % 64 == 0 and ly.x.C <= fuse_max_cin and FUSE_OFFSET
                        and not (small_unfuse and ly.main in slot_cps) and ly.main not in unfuse_slots)
            ly.nkk = 2 if (c == 1 or ly.x.C % 64) else nkk
            ly.splits = max(1, nchunks // c)
            om = part = None
            raw = False
            if not ly.fused and split_offsets == 2 and ly.x.C % 64 == 0 and pk['w_off_wino'] is not None:
                # raw sums by a conv launch of its own, free to use the Winograd shapes (2.25x fewer MFMAs): the bias
                # and the mask sigmoid are applied by the DCN launch
                raw = True
                omv = ops.new_view(N, ly.x.H, ly.x.W, 32, dev)
                part = omv.buf
                d = ops.make_conv_desc(ly.x, pk['w_off'], 27, 3, 1, out=omv, w_wino=pk['w_off_wino'])
                if tune:
                    autotune.tune_conv(d, dev)
                convs.setdefault(ly.main, []).append(
                    _Launch(ly.name + '.offset', 'conv', d, (ly.x, omv, pk),
                            ws_need=lib.ct_conv2d_workspace_bytes(ctypes.byref(d))))
            elif not ly.fused and split_offsets and ly.x.C % 64 == 0:
                part = torch.empty((ly.x.C // 64) * N * ly.x.H * ly.x.W * 32, dtype=torch.float32, device=dev)
                offs.setdefault(ly.main, []).append(ly)
            elif not ly.fused:
                om = ops.new_view(N, ly.x.H, ly.x.W, 32, dev)
                d = ops.make_conv_desc(ly.x, pk['w_off'], 27, 3, 1, shift=pk['b_off'], out=om, sig=(18, 27))
                if tune:
                    autotune.tune_conv(d, dev)
                convs.setdefault(ly.main, []).append(
                    _Launch(ly.name + '.offset', 'conv', d, (ly.x, om, pk),
                            ws_need=lib.ct_conv2d_workspace_bytes(ctypes.byref(d))))
            own = ly.fused or part is not None
            dd = ops.make_dcn_desc(ly.x, om, pk['w'], ly.cout, pk['scale'], pk['shift'], True, ly.out, up=ly.up,
                                   split_k=ly.splits, algo=3264, om_partial=part, raw_offsets=raw,
                                   w_off=pk['w_off'] if own else None, b_off=pk['b_off'] if own else None,
                                   # split_offsets == 3 (round 6): the slot's K-split OFFSETS launch in its Winograd form
                                   w_off_wino=pk['w_off_wino'] if (split_offsets == 3 and not ly.fused) else None)
            need_c = ctypes.c_size_t(0)
            _lib.check(lib.ct_dcn_v2_group_plan(ctypes.byref(dd), ctypes.byref(need_c), None), 'DCN layer ' + ly.name)
            need = need_c.value
            ly.use_ws = need > 0
            keep = [ly.x, om, part, ly.out, pk, ly.up]
            if need:
                ws = torch.empty(need // 4, dtype=torch.float32, device=dev)
                dd.workspace, dd.workspace_bytes = ws.data_ptr(), need
                keep.append(ws)
            ly.desc = (dd, keep)
            if ly.use_ws:
                ly.finish = max(ly.main, (t_of(ly.up[2]) + 1) // 2) if ly.up is not None else ly.main
                done = 2 * ly.finish + 1
            else:
                ly.finish = None
                done = 2 * ly.main
            time_of[id((ly.up[3] if ly.up is not None else ly.out).buf)] = done
            slots.setdefault(ly.main, {'main': [], 'finish': []})['main'].append(ly)
            if ly.finish is not None:
                slots.setdefault(ly.finish, {'main': [], 'finish': []})['finish'].append(ly)
        out = []

        def group(lys, phases, tag):
            if phases == _lib.CT_DCN_MAIN:
                # longest workgroups first (round 6): the dispatcher hands out workgroups in id order, so the layer whose
                # workgroups run the most (chunk, tap) steps starts first and the short ones fill the tail of the launch
                lys = sorted(lys, key=lambda ly: -((ly.x.C // 32 + ly.splits - 1) // ly.splits))
            for i in range(0, len(lys), 4):
                part = lys[i:i + 4]
                arr = (_lib.DcnDesc * len(part))()
                keep = []
                def can_persist(ly):
                    cpsplit = -(-(ly.x.C // 32) // ly.splits)
                    return (not ly.fused and (ly.x.C // 32) % cpsplit == 0
                            and cpsplit % (4 if ly.nkk == 4 else 2) == 0)
                pers = bool(persist) and phases == _lib.CT_DCN_MAIN and all(can_persist(ly) for ly in part)
                for j, ly in enumerate(part):
                    ctypes.memmove(ctypes.byref(arr[j]), ctypes.byref(ly.desc[0]), ctypes.sizeof(_lib.DcnDesc))
                    arr[j].algo = (43264 if ly.nkk == 4 else 3264) + (20000 if pers and ly.nkk == 4 else 50000 if pers else 0)
                    keep.append(ly.desc[1])
                name = '%s[%s]' % (tag, ' + '.join(ly.name for ly in part))
                out.append(_Launch(name, 'dcn_group', (arr, len(part), phases), keep))

        for t in sorted(slots):
            out.extend(convs.get(t, []))
            if offs.get(t):
                group(offs[t], _lib.CT_DCN_OFFSETS, 'dcn.offsets')
            lys = slots[t]['main']
            if lys:
                group(lys, _lib.CT_DCN_MAIN, 'dcn')
            if slots[t]['finish']:
                group(slots[t]['finish'], _lib.CT_DCN_FINISH, 'dcn.finish')
        return out

    @staticmethod
    def _dcn_slot_sizes(layers, produced0, N, cps, with_mains=False):
        """{MAIN slot: workgroups} of the schedule with ``cps`` chunks per split everywhere.  A split layer is readable
        one time step later, which rounds to the same next MAIN slot for a consumer that READS it; only a `proj` node
        whose SKIP input comes from a split layer finishes one slot later (at 512x512: ida_up.node_1/2 move from slots
        7/8 to 8/10 when the 64-channel nodes are split, cps = 1)"""
        time_of = dict(produced0)
        sizes = {}
        mains = []
        for ly in layers:
            main = time_of.get(id(ly.x.buf), 0) // 2 + 1
            mains.append(main)
            splits = max(1, (ly.x.C // 32) // cps)
            use_ws = splits > 1 or ly.up is not None
            if use_ws:
                fin = max(main, (time_of.get(id(ly.up[2].buf), 0) + 1) // 2) if ly.up is not None else main
                done = 2 * fin + 1
            else:
                done = 2 * main
            time_of[id((ly.up[3] if ly.up is not None else ly.out).buf)] = done
            tiles = N * ((ly.x.H + 1) // 2) * ((ly.x.W + 15) // 16) * ((ly.cout + 63) // 64)
            sizes[main] = sizes.get(main, 0) + tiles * splits
        return (sizes, mains) if with_mains else sizes

    def _time_launches(self, launches, reps=10):
        """device time (us) of a launch list, by graph replay on a side stream"""
        plan = {'launches': launches}
        need = max([l.ws_need for l in launches if l.fn == 'conv'] + [16])
        dev = next(self.buffers()).device
        ws = torch.empty(need // 4, dtype=torch.float32, device=dev)
        for l in launches:
            if l.fn == 'conv':
                l.args.workspace, l.args.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._run_plan(plan)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with _lib.capture_guard(collect=False), torch.cuda.graph(g):
            for _ in range(reps):
                self._run_plan(plan)
        g.replay()
        torch.cuda.synchronize()
        best = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / reps
            best = t if best is None else min(best, t)
        del g
        return best

    def _tune_dcn_schedule(self, layers, produced0, N, H, W, dev, tune):
        """Pick the knobs of _schedule_dcn for this (batch, size) by timing the whole 16-node sequence (a HIP graph of
        exactly its launches) for every candidate; the choice is cached like the per-layer ones (pinned table /
        CENTERTRACK_TUNE_CACHE) because it fixes the fp32 summation order.  CENTERTRACK_DCN_KNOBS="a,b,c" forces one."""
        env = os.environ.get('CENTERTRACK_DCN_KNOBS', '')
        if env:
            knobs = tuple(int(v) for v in env.split(','))
            return knobs, self._schedule_dcn(layers, produced0, N, dev, knobs, tune)
        default = (128, 4, 2, 1, 0, 0, 0)
        if not tune:
            return default, self._schedule_dcn(layers, produced0, N, dev, default, tune)
        key = 'dcnplan5:%d,%d,%d' % (N, H, W)
        key4 = 'dcnplan4:%d,%d,%d' % (N, H, W)          # (tables before the persistent launches: six knobs)
        key3 = 'dcnplan3:%d,%d,%d' % (N, H, W)          # (round-2 tables: four knobs, no fine-split slots)
        autotune._load_file()
        retune = os.environ.get('CENTERTRACK_DCN_RETUNE', '0') == '1'      # (tools/retune_dcn.py: measure again)
        if key in autotune._CACHE and not retune:
            knobs = tuple(int(v) for v in autotune._CACHE[key][:-1])
            return knobs, self._schedule_dcn(layers, produced0, N, dev, knobs)
        if key4 in autotune._CACHE and not retune:
            knobs = tuple(int(v) for v in autotune._CACHE[key4][:-1]) + (0,)
            return knobs, self._schedule_dcn(layers, produced0, N, dev, knobs)
        if key3 in autotune._CACHE and not retune:
            knobs = tuple(int(v) for v in autotune._CACHE[key3][:4]) + (0, 0, 0)
            return knobs, self._schedule_dcn(layers, produced0, N, dev, knobs)
        best = None
        tried = []
        # (split_offsets = 0 -- one conv launch per un-fused layer -- never won in round 2's sweeps: 368 against 336 us
        # at one stream, 1932 against 1818 us at eight; it stays reachable through CENTERTRACK_DCN_KNOBS.  Neither did
        # 128-cout tiles for the layers with >= 128 couts in a MAIN launch of their own: 6866 against 6830 us at 32
        # streams, 1826 against 1806 at eight -- the second launch per slot costs what the wider tile gains)
        for fuse_max in (0, 64, 128, 256):
            for cps in (2, 4, 8):
                for nkk in (2, 4):
                    for so in ((1, 2, 3) if fuse_max < 256 else (1,)):
                        small = [t for t, w in self._dcn_slot_sizes(layers, produced0, N, cps).items() if w < SMALL_SLOT_WGS]
                        for cs, un in (((0, 0), (0, 2)) + tuple((c, u) for c in (2, 1) if c < cps for u in (0, 1, 2)) if small
                                       else ((0, 0), (0, 2))):
                            knobs = (fuse_max, cps, nkk, so, cs, un, 0)
                            launches = self._schedule_dcn(layers, produced0, N, dev, knobs)
                            us = self._time_launches(launches)
                            if os.environ.get('CENTERTRACK_TUNE_VERBOSE'):
                                print('dcn schedule N=%d %dx%d knobs %s: %d launches, %.1f us' % (N, H, W, knobs, len(launches), us))
                            tried.append((us, knobs))
                            if best is None or us < best[0]:
                                best = (us, knobs)
                            del launches
        # second stage (CENTERTRACK_DCN_TUNE_PERSIST=1): the persistent MAIN launches on top of the five best schedules and of
        # every schedule that keeps all offset convs out of the MAIN launches (the only layers a persistent launch takes).
        # Off by default: over the 28 pinned shapes the persistent form lost everywhere, by 1 .. 8 %
        # (profiles/r06_z_persistent_vs_per_tile_all_shapes.txt)
        tried.sort()
        stage2 = os.environ.get('CENTERTRACK_DCN_TUNE_PERSIST', '0') == '1'
        for us0, k0 in (tried[:5] + [t for t in tried[5:] if t[1][0] == 0 and t[1][5] == 0]) if stage2 else ():
            knobs = k0[:6] + (1,)
            launches = self._schedule_dcn(layers, produced0, N, dev, knobs)
            if any(l.fn == 'dcn_group' and int(l.args[0][0].algo) >= 50000 for l in launches):
                us = self._time_launches(launches)
                if os.environ.get('CENTERTRACK_TUNE_VERBOSE'):
                    print('dcn schedule N=%d %dx%d knobs %s: %d launches, %.1f us (%.1f without)' % (N, H, W, knobs, len(launches), us, us0))
                if us < best[0]:
                    best = (us, knobs)
            del launches
        autotune._CACHE[key] = tuple(best[1]) + (round(best[0], 1),)
        autotune._save_file()
        return best[1], self._schedule_dcn(layers, produced0, N, dev, best[1])

    @staticmethod
    def plan_signature(plan):
        """Everything of a plan that fixes the fp32 summation order, as one string: per launch its name, kernel choice
        (algo) and split-K, the per-layer DCN tile / split / offset mode, the DCN schedule knobs.  Ranks of a multi-GPU
        job compare its hash (parallel.check_same_plan) before they start."""
        parts = ['dcn_knobs=%s' % (tuple(plan.get('dcn_knobs', ())),)]
        for l in plan['launches']:
            if l.fn == 'conv':
                parts.append('%s:conv:%d:%d' % (l.name, int(l.args.algo), int(l.args.split_k)))
            elif l.fn == 'dcn_group':
                arr, n, phases = l.args
                parts.append('%s:dcn:%d:%s' % (l.name, phases, ','.join(
                    '%d/%d/%d' % (int(arr[j].algo), int(arr[j].split_k), int(arr[j].fuse_offset)) for j in range(n))))
            else:
                parts.append('%s:%s' % (l.name, l.fn))
        return ';'.join(parts)

    def get_plan(self, N, H, W, with_img, with_hm, fuse_sigmoid=False, sparse_heads=False):
        key = (N, H, W, with_img, with_hm, fuse_sigmoid, sparse_heads)
        if key not in self._plans:
            self._plans[key] = self._build_plan(*key)
        return self._plans[key]

    def forward_plan(self, plan, x, pre_img=None, pre_hm=None):
        """Copy the inputs into the plan's static buffers and enqueue all launches."""
        xi, ii, hi = plan['inputs']
        xi.copy_(x)
        if ii is not None:
            ii.copy_(pre_img)
        if hi is not None:
            hi.copy_(pre_hm)
        self._run_plan(plan)
        return plan['outputs']

    @torch.no_grad()
    def forward(self, x, pre_img=None, pre_hm=None, fuse_sigmoid=False):
        """Reference signature (base_model.py:73): returns ``[ {head: tensor[B,c,h,w]} ]``
        (raw logits unless ``fuse_sigmoid``, in which case hm / dep carry the transforms of
        ``Detector._sigmoid_output``, detector.py:300-308).  Outputs are fresh tensors."""
        if pre_img is not None and not self.pre_img:
            raise _lib.CTError('model was built with pre_img=False')
        if pre_hm is not None and not self.pre_hm:
            raise _lib.CTError('model was built with pre_hm=False')
        N, _, H, W = x.shape
        plan = self.get_plan(N, H, W, pre_img is not None, pre_hm is not None, fuse_sigmoid)
        out = self.forward_plan(plan, x, pre_img, pre_hm)
        z = OrderedDict((k, v.clone()) for k, v in out.items())
        if self.model_output_list:
            return [[z[h] for h in sorted(self.heads)]]
        return [z]


def create_model(arch, head, head_conv, opt=None):
    """reference model.py:24-29; only the published architecture ``dla_34`` exists here."""
    if arch != 'dla_34':
        raise ValueError('centertrack_amd implements arch "dla_34" only (got %r)' % arch)
    if opt is not None and getattr(opt, 'dla_node', 'dcn') != 'dcn':
        raise ValueError('only dla_node="dcn" is implemented')
    if opt is not None and getattr(opt, 'head_kernel', 3) != 3:
        raise ValueError('only head_kernel=3 is implemented')
    return DLASegHIP(head, head_conv,
                     pre_img=getattr(opt, 'pre_img', True) if opt is not None else True,
                     pre_hm=getattr(opt, 'pre_hm', True) if opt is not None else True,
                     depth_scale=getattr(opt, 'depth_scale', 1.0) if opt is not None else 1.0,
                     model_output_list=getattr(opt, 'model_output_list', False) if opt is not None else False)


def load_model(model, model_path, opt=None, optimizer=None):
    """reference model.py:31-90 (inference part): ``{'epoch', 'state_dict'}`` checkpoint, ``module.`` prefix
    stripped, unknown keys dropped, missing keys keep their initialisation; a parameter whose shape differs -- or,
    under ``opt.reset_hm``, an ``hm*`` parameter with 80 / 1 rows -- is skipped, or with ``opt.reuse_hm`` cut /
    padded along its first axis (a COCO-pretrained heat-map head reused for another class count, model.py:49-63)."""
    ckpt = torch.load(model_path, map_location='cpu')
    sd_in = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    if 'epoch' in ckpt:
        print('loaded {}, epoch {}'.format(model_path, ckpt['epoch']))
    sd = {}
    for k, v in sd_in.items():
        sd[k[7:] if k.startswith('module') and not k.startswith('module_list') else k] = v
    own = model.state_dict()
    reset_hm = bool(getattr(opt, 'reset_hm', False))
    reuse_hm = bool(getattr(opt, 'reuse_hm', False))
    keep = {}
    for k, v in sd.items():
        if k in own:
            mismatch = tuple(v.shape) != tuple(own[k].shape)
            if mismatch or (reset_hm and k.startswith('hm') and v.shape[0] in (80, 1)):
                if reuse_hm and tuple(v.shape[1:]) == tuple(own[k].shape[1:]):
                    print('Reusing parameter {}, required shape{}, loaded shape{}.'.format(
                        k, tuple(own[k].shape), tuple(v.shape)))
                    n = min(v.shape[0], own[k].shape[0])      # (the reference only ever cuts: `a < a` is never true)
                    merged = own[k].clone()
                    merged[:n] = v[:n]
                    keep[k] = merged
                else:
                    print('Skip loading parameter {}, required shape{}, loaded shape{}.'.format(
                        k, tuple(own[k].shape), tuple(v.shape)))
            else:
                keep[k] = v
        else:
            print('Drop parameter {}.'.format(k))
    for k in own:
        if k not in keep:
            print('No param {}.'.format(k))
    model.load_state_dict(keep, strict=False)
    return model
