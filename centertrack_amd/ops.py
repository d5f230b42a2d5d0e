"""Tensor-level wrappers over the C ABI (torch is only the allocator / stream owner).

``View`` = an NHWC fp32 activation living in (a channel slice of) a contiguous
``[N,H,W,ld]`` buffer; every op reads / writes views so concatenations
(reference ``Root.forward``, dla.py:164-166) never copy."""
import ctypes

import torch

from . import _lib
from ._lib import CT_OUT_NCHW, CT_RELU, ConvDesc, DcnDesc, DecodeDesc


class View(object):
    __slots__ = ('buf', 'c0', 'C')

    def __init__(self, buf, c0=0, C=None):
        assert buf.dim() == 4 and buf.is_contiguous() and buf.dtype == torch.float32
        self.buf = buf
        self.c0 = c0
        self.C = buf.shape[3] - c0 if C is None else C
        assert 0 <= c0 and c0 + self.C <= buf.shape[3]

    N = property(lambda s: s.buf.shape[0])
    H = property(lambda s: s.buf.shape[1])
    W = property(lambda s: s.buf.shape[2])
    ld = property(lambda s: s.buf.shape[3])
    ptr = property(lambda s: s.buf.data_ptr() + 4 * s.c0)

    def slice(self, c0, C):
        return View(self.buf, self.c0 + c0, C)

    def to_nchw(self):
        """torch view (no copy) of the slice as [N,C,H,W] -- for tests / debugging."""
        return self.buf[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2)


def new_view(N, H, W, C, device, ld=None):
    return View(torch.empty((N, H, W, ld or C), dtype=torch.float32, device=device), 0, C)


def view_from_nchw(x):
    """NCHW torch tensor -> fresh NHWC view via the HIP converter."""
    N, C, H, W = x.shape
    x = x.contiguous().float()
    ld = (C + 3) // 4 * 4
    v = View(torch.zeros((N, H, W, ld), dtype=torch.float32, device=x.device), 0, C)
    _lib.check(_lib.load().ct_nchw_to_nhwc(x.data_ptr(), N, C, H, W, v.ptr, v.ld, _lib.stream_ptr()),
               'ct_nchw_to_nhwc')
    return v


def view_to_nchw(v):
    out = torch.empty((v.N, v.C, v.H, v.W), dtype=torch.float32, device=v.buf.device)
    _lib.check(_lib.load().ct_nhwc_to_nchw(v.ptr, v.N, v.C, v.H, v.W, v.ld, out.data_ptr(), _lib.stream_ptr()),
               'ct_nhwc_to_nchw')
    return out


def pack_weight(w):
    """OIHW conv weight (cuda, fp32) -> MFMA fragment layout (see centertrack_hip.h)."""
    lib = _lib.load()
    w = w.contiguous().float()
    Cout, Cin, ks, ks2 = w.shape
    assert ks == ks2 and Cin % 16 == 0, 'Cin must be a multiple of 16'
    out = torch.empty(lib.ct_packed_weight_elems(Cout, Cin, ks), dtype=torch.float32, device=w.device)
    _lib.check(lib.ct_pack_conv_weight(w.data_ptr(), out.data_ptr(), Cout, Cin, ks, _lib.stream_ptr()),
               'ct_pack_conv_weight')
    return out


def pack_winograd(w):
    """OIHW 3x3 conv weight -> Winograd F(2x2,3x3) fragments (ct_conv2d algo 201 / 202)"""
    lib = _lib.load()
    w = w.contiguous().float()
    Cout, Cin, ks, ks2 = w.shape
    assert ks == 3 and ks2 == 3 and Cin % 16 == 0
    out = torch.empty(lib.ct_packed_winograd_elems(Cout, Cin), dtype=torch.float32, device=w.device)
    _lib.check(lib.ct_pack_winograd_weight(w.data_ptr(), out.data_ptr(), Cout, Cin, _lib.stream_ptr()),
               'ct_pack_winograd_weight')
    return out


def _p(t):
    return None if t is None else t.data_ptr()


def make_conv_desc(x, wp, Cout, ks, stride=1, scale=None, shift=None, res=None, relu=False, out=None,
                   out_nchw=None, sig=(0, 0), dep=(0, 0), depth_scale=1.0, workspace=None, split_k=0, algo=0,
                   w_wino=None, pool=None, proj=None):
    """``pool``: NHWC view [N, H/2, W/2, Cin] that receives the 2x2 max-pool of ``x`` as a side output (3x3 stride-2);
    ``proj`` = (packed [Cout, Cin, 1, 1] weight, scale, shift, NHWC output view): Tree.project of the pooled input
    (conv1x1 + BN, no ReLU) as a second output of the same launch"""
    d = ConvDesc()
    d.x, d.N, d.H, d.W, d.Cin, d.ldx = x.ptr, x.N, x.H, x.W, x.C, x.ld
    d.w_packed, d.Cout, d.ks, d.stride = wp.data_ptr(), Cout, ks, stride
    d.scale, d.shift = _p(scale), _p(shift)
    if res is not None:
        d.res, d.ldr = res.ptr, res.ld
    flags = CT_RELU if relu else 0
    if out_nchw is not None:
        assert out is None and out_nchw.is_contiguous()
        d.y, d.ldy = out_nchw.data_ptr(), 0
        flags |= CT_OUT_NCHW
    else:
        d.y, d.ldy = out.ptr, out.ld
    d.flags = flags
    d.sig_lo, d.sig_hi = sig
    d.dep_lo, d.dep_hi = dep
    d.depth_scale = depth_scale
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    d.split_k = split_k
    d.algo = algo
    if w_wino is not None:
        d.w_winograd = w_wino.data_ptr()
    if pool is not None:
        assert ks == 3 and stride == 2 and (pool.N, pool.H, pool.W, pool.C) == (x.N, x.H // 2, x.W // 2, x.C)
        d.pool_y, d.pool_ld = pool.ptr, pool.ld
    if proj is not None:
        pw, psc, psh, py = proj
        assert ks == 3 and stride == 2 and (py.N, py.H, py.W, py.C) == (x.N, x.H // 2, x.W // 2, Cout)
        d.proj_w_packed, d.proj_scale, d.proj_shift = pw.data_ptr(), _p(psc), _p(psh)
        d.proj_y, d.proj_ldy = py.ptr, py.ld
    return d


def conv2d(x, wp, Cout, ks, stride=1, out=None, **kw):
    """y = act(conv(x) * scale + shift + res); allocates the output view if not given."""
    lib = _lib.load()
    pad = ks // 2
    Ho = (x.H + 2 * pad - ks) // stride + 1
    Wo = (x.W + 2 * pad - ks) // stride + 1
    if out is None and kw.get('out_nchw') is None:
        out = new_view(x.N, Ho, Wo, Cout, x.buf.device, ld=(Cout + 3) // 4 * 4)
    d = make_conv_desc(x, wp, Cout, ks, stride, out=out, **kw)
    if d.workspace is None and kw.get('split_k', 0) != 1:
        need = lib.ct_conv2d_workspace_bytes(ctypes.byref(d))
        if need:
            ws = torch.empty(need // 4, dtype=torch.float32, device=x.buf.device)
            d.workspace, d.workspace_bytes = ws.data_ptr(), need
            kw['_keep'] = ws
    _lib.check(lib.ct_conv2d(ctypes.byref(d), _lib.stream_ptr()), 'ct_conv2d')
    return out if out is not None else kw['out_nchw']


def make_dcn_desc(x, om, wp, Cout, scale, shift, relu, out, workspace=None, split_k=0, algo=0, w_off=None,
                  b_off=None, up=None, om_partial=None, raw_offsets=False, w_off_wino=None):
    """``w_off`` (packed conv_offset_mask weight) + ``b_off`` given: the offset/mask conv runs inside the DCN
    launch (``om`` may be None) -- or, with ``om_partial`` (a float buffer of ct_dcn_v2_offsets_bytes), K-split by
    the CT_DCN_OFFSETS launch; otherwise ``om`` is the precomputed NHWC offset/mask map."""
    d = DcnDesc()
    d.x, d.N, d.H, d.W, d.Cin, d.ldx = x.ptr, x.N, x.H, x.W, x.C, x.ld
    if om is not None:
        d.om, d.ldom = om.ptr, om.ld
    if w_off is not None:
        d.fuse_offset, d.w_off_packed, d.b_off = 1, w_off.data_ptr(), b_off.data_ptr()
        if om_partial is not None:
            d.fuse_offset = 3 if raw_offsets else 2          # (3: another launch writes the raw sums into om_partial)
            d.om_partial, d.om_partial_bytes = om_partial.data_ptr(), om_partial.numel() * om_partial.element_size()
            if w_off_wino is not None and not raw_offsets:    # (the OFFSETS launch runs this layer's chunks as Winograd tiles)
                d.w_off_winograd = w_off_wino.data_ptr()
    if up is not None:                 # (upsample_weight [4f^2,C], f, skip view, output view): fused IDAUp step
        w_up, f, skip, up_out = up
        d.up_w, d.up_f, d.up_skip, d.up_lds = w_up.data_ptr(), f, skip.ptr, skip.ld
        d.up_y, d.up_ldy = up_out.ptr, up_out.ld
    d.w_packed, d.Cout = wp.data_ptr(), Cout
    d.scale, d.shift = _p(scale), _p(shift)
    d.y, d.ldy = out.ptr, out.ld
    d.flags = CT_RELU if relu else 0
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    d.split_k = split_k
    d.algo = algo
    return d


def dcn_v2(x, om, wp, Cout, scale=None, shift=None, relu=False, out=None, split_k=0, algo=0, w_off=None, b_off=None,
           up=None, split_offsets=False, w_off_wino=None):
    lib = _lib.load()
    if out is None:
        out = new_view(x.N, x.H, x.W, Cout, x.buf.device)
    part = None
    if split_offsets:
        part = torch.empty((x.C // 64) * x.N * x.H * x.W * 32, dtype=torch.float32, device=x.buf.device)
    d = make_dcn_desc(x, om, wp, Cout, scale, shift, relu, out, split_k=split_k, algo=algo, w_off=w_off, b_off=b_off,
                      up=up, om_partial=part, w_off_wino=w_off_wino if split_offsets else None)
    ws = None
    if split_k != 1:
        need = lib.ct_dcn_v2_workspace_bytes(ctypes.byref(d))
        if need:
            ws = torch.empty(need // 4, dtype=torch.float32, device=x.buf.device)
            d.workspace, d.workspace_bytes = ws.data_ptr(), need
    _lib.check(lib.ct_dcn_v2(ctypes.byref(d), _lib.stream_ptr()), 'ct_dcn_v2')
    return out


def stem(x, pre_img, pre_hm, w_x, w_img, w_hm, scale3, shift3, out=None):
    N, _, H, W = x.shape
    if out is None:
        out = new_view(N, H, W, 16, x.device)
    _lib.check(_lib.load().ct_stem_forward(
        x.data_ptr(), _p(pre_img), _p(pre_hm), N, H, W, w_x.data_ptr(), _p(w_img), _p(w_hm),
        scale3.data_ptr(), shift3.data_ptr(), out.ptr, out.ld, _lib.stream_ptr()), 'ct_stem_forward')
    return out


def maxpool2x2(x, out=None):
    if out is None:
        out = new_view(x.N, x.H // 2, x.W // 2, x.C, x.buf.device)
    _lib.check(_lib.load().ct_maxpool2x2(x.ptr, x.N, x.H, x.W, x.C, x.ld, out.ptr, out.ld, _lib.stream_ptr()),
               'ct_maxpool2x2')
    return out


def upsample_weight(w):
    """ConvTranspose2d depth-wise weight [C,1,2f,2f] -> the kernel's [2f,2f,C] layout"""
    return w.reshape(w.shape[0], -1).t().contiguous()


def upsample_add(x, w, f, skip, out=None):
    """w: the module weight [C,1,2f,2f] (transposed here) or an ``upsample_weight`` result [4f^2,C]"""
    if w.dim() == 4:
        w = upsample_weight(w)
    if out is None:
        out = new_view(x.N, x.H * f, x.W * f, x.C, x.buf.device)
    _lib.check(_lib.load().ct_upsample_add(x.ptr, x.N, x.H, x.W, x.C, x.ld, w.data_ptr(), f, skip.ptr, skip.ld,
                                           out.ptr, out.ld, _lib.stream_ptr()), 'ct_upsample_add')
    return out


# field order of a packed decode row after (score, cls, xs0, ys0)
_DECODE_REST = ['tracking', 'dep', 'rot', 'dim', 'amodel_offset', 'nuscenes_att', 'velocity']


def decode_layout(head_names):
    """[(field, start, width)] of one packed row, mirroring ct_decode_row_floats."""
    names = set(head_names)
    lay = [('scores', 0, 1), ('clses', 1, 1), ('xs', 2, 1), ('ys', 3, 1)]
    f = 4
    if names & {'wh', 'ltrb', 'ltrb_amodal'}:
        lay.append(('bboxes', f, 4)); f += 4
    if 'ltrb_amodal' in names:
        lay.append(('bboxes_amodal', f, 4)); f += 4
    for n in _DECODE_REST:
        if n in names:
            lay.append((n, f, _lib.HEAD_CH[n])); f += _lib.HEAD_CH[n]
    return lay, f


class Decoder(object):
    """Pre-built ct_decode call (+ ct_decode_pose when the pose heads ``hps`` / ``hm_hp`` are given) for fixed
    shapes (graph-capture friendly).  One packed buffer ``out`` [B,K,F]: the rows of ct_decode followed, for the
    pose task, by the refined key points [2J] and ``kps_score`` [1]."""

    @staticmethod
    def row_floats(heads, sparse=None):
        """floats per packed row for this set of heads (the pose fields included; ``sparse``: names of sparse heads)"""
        _, F = decode_layout([n for n in heads if n in _lib.HEAD_INDEX] + list(sparse or ()))
        if 'hps' in heads and 'hm_hp' in heads:
            F += heads['hps'].shape[1] + 1
        return F

    def __init__(self, hm, heads, K, host_out=None, done_flag=None, sparse=None):
        """``host_out`` (pinned host float32 [B,K,F]) / ``done_flag`` (pinned host int32): the decode also stores the
        rows straight into host memory and raises the flag when they are complete (no pose heads).
        ``sparse`` (round 5, opt-in): ``{'feat': NHWC view of the 64-channel feature map, 'heads': [(name, w1_packed, b1,
        w2 [c,256], b2)], 'depth_scale', 'zero_tracking'}`` -- these regression heads have no map in ``heads``; they are
        evaluated at the K winners only (ct_sparse_heads_desc)"""
        lib = _lib.load()
        self.K = K
        B, C, h, w = hm.shape
        self.hm, self.heads = hm, dict(heads)

        def planes_ok(t):      # every image's [c,h,w] block contiguous; images may be strided (channel slices)
            return t.stride(3) == 1 and t.stride(2) == w and (t.shape[1] == 1 or t.stride(1) == h * w)
        d = DecodeDesc()
        assert planes_ok(hm)
        d.hm, d.B, d.C, d.h, d.w, d.K = hm.data_ptr(), B, C, h, w, K
        d.hm_batch_stride = hm.stride(0)
        for name, t in heads.items():
            if name in _lib.HEAD_INDEX:
                assert planes_ok(t) and t.shape[1] == _lib.HEAD_CH[name], name
                d.heads[_lib.HEAD_INDEX[name]] = t.data_ptr()
                d.head_batch_stride[_lib.HEAD_INDEX[name]] = t.stride(0)
        self.sparse = None
        if sparse is not None:
            assert 'hps' not in heads and all(n not in heads for n, *_ in sparse['heads'])
            sp = _lib.SparseHeadsDesc()
            feat = sparse['feat']
            flip = bool(sparse.get('flip'))
            assert feat.C == 64 and (feat.N, feat.H, feat.W) == ((2 * B if flip else B), h, w)
            sp.flip_B = B if flip else 0
            sp.feat, sp.ldf, sp.nheads = feat.ptr, feat.ld, len(sparse['heads'])
            for i, (name, w1, b1, w2, b2) in enumerate(sparse['heads']):
                assert tuple(w2.shape) == (_lib.HEAD_CH[name], 256) and w2.is_contiguous() and b1.numel() == 256
                sp.head[i] = _lib.HEAD_INDEX[name]
                sp.flip_mode[i] = (1 if name in ('wh', 'dep', 'dim') else 2 if name == 'amodel_offset' else 0) if flip else 0
                sp.w1[i], sp.b1[i], sp.w2[i], sp.b2[i] = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
            sp.depth_scale = float(sparse.get('depth_scale', 1.0))
            sp.zero_tracking = int(bool(sparse.get('zero_tracking', False)))
            self.sparse = (sp, sparse)                     # (keeps the tensors alive)
            d.sparse = ctypes.pointer(sp)
        self.layout, self.F = decode_layout([n for n in heads if n in _lib.HEAD_INDEX] +
                                            ([n for n, *_ in sparse['heads']] if sparse is not None else []))
        F0 = self.F
        assert lib.ct_decode_row_floats(ctypes.byref(d)) == F0
        self.pose = None
        if 'hps' in heads:                                 # pose branch, decode.py:161-171
            J = heads['hps'].shape[1] // 2
            if 'hm_hp' in heads:
                self.layout = self.layout + [('hps', F0, 2 * J), ('kps_score', F0 + 2 * J, 1)]
                self.F = F0 + 2 * J + 1
            else:                                          # decode.py:80-81: no refinement, kps_score = kps
                raise _lib.CTError('the pose branch without an hm_hp head is not implemented')
        self.out = torch.empty((B, K, self.F), dtype=torch.float32, device=hm.device)
        self.inds = torch.empty((B, K), dtype=torch.int64, device=hm.device)
        nbytes = lib.ct_decode_workspace_bytes(ctypes.byref(d))
        if nbytes == 0:
            _lib.check(1, 'ct_decode_workspace_bytes')
        # (zeroed: the block holds a reserved counter area besides the keys -- nothing in it is ever read uninitialised)
        self.ws = torch.zeros(nbytes // 8 + 1, dtype=torch.int64, device=hm.device)
        d.out, d.inds = self.out.data_ptr(), self.inds.data_ptr()
        d.out_stride = self.F
        d.workspace, d.workspace_bytes = self.ws.data_ptr(), nbytes
        self.direct = False
        if host_out is not None and done_flag is not None and 'hps' not in heads:
            assert tuple(host_out.shape) == (B, K, self.F) and host_out.is_pinned() and done_flag.is_pinned()
            self.done_counter = torch.zeros((4,), dtype=torch.int32, device=hm.device)
            d.host_out, d.done_flag = host_out.data_ptr(), done_flag.data_ptr()
            d.done_counter = self.done_counter.data_ptr()
            self.direct = True
        self.desc = d
        if 'hps' in heads:
            hps, hm_hp = heads['hps'], heads['hm_hp']
            off = heads.get('hp_offset', heads.get('reg'))
            assert planes_ok(hps) and hm_hp.is_contiguous() and hm_hp.shape[1] == J and (off is None or planes_ok(off))
            pd = _lib.PoseDesc()
            pd.rows, pd.row_floats, pd.inds = self.out.data_ptr(), self.F, self.inds.data_ptr()
            # the gate box of _update_kps_with_hm is generic_decode's local `bboxes`: wh, overridden by ltrb -- never the
            # ltrb_amodal box that replaces ret['bboxes'] in the packed row (decode.py:123,137 vs 159)
            if 'ltrb_amodal' in heads or not ({'wh', 'ltrb'} & set(heads)):
                pd.box_col = -1
                for name, fld in (('wh', 'box_wh'), ('ltrb', 'box_ltrb')):
                    if name in heads:
                        assert planes_ok(heads[name])
                        setattr(pd, fld, heads[name].data_ptr())
                        setattr(pd, fld + '_batch_stride', heads[name].stride(0))
                if 'wh' in heads and 'reg' in heads:
                    pd.box_reg, pd.box_reg_batch_stride = heads['reg'].data_ptr(), heads['reg'].stride(0)
            else:
                pd.box_col = 4
            pd.B, pd.h, pd.w, pd.K, pd.num_joints = B, h, w, K, J
            pd.hps, pd.hm_hp, pd.hp_offset = hps.data_ptr(), hm_hp.data_ptr(), _p(off)
            pd.hps_batch_stride = hps.stride(0)
            pd.hp_offset_batch_stride = off.stride(0) if off is not None else 0
            pd.out, pd.out_stride = self.out.data_ptr() + 4 * F0, self.F
            pbytes = lib.ct_decode_pose_workspace_bytes(ctypes.byref(pd))
            if pbytes == 0:
                _lib.check(1, 'ct_decode_pose_workspace_bytes')
            self.pose_ws = torch.empty(pbytes // 8 + 1, dtype=torch.int64, device=hm.device)
            pd.workspace, pd.workspace_bytes = self.pose_ws.data_ptr(), self.pose_ws.numel() * 8
            self.pose = pd

    def run(self):
        _lib.check(_lib.load().ct_decode(ctypes.byref(self.desc), _lib.stream_ptr()), 'ct_decode')
        if self.pose is not None:
            _lib.check(_lib.load().ct_decode_pose(ctypes.byref(self.pose), _lib.stream_ptr()), 'ct_decode_pose')
        return self.out

    def unpack(self, packed):
        """packed [B,K,F] (torch or numpy) -> the reference's ``dets`` dict (decode.py:99-180)."""
        ret = {}
        for name, s, wd in self.layout:
            v = packed[..., s:s + wd]
            ret[name] = v[..., 0] if name in ('scores', 'clses', 'xs', 'ys', 'kps_score') else v
        ret['cts'] = packed[..., 2:4]
        return ret
