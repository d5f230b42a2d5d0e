#!/usr/bin/env python
"""Micro-benchmark of the conv / DCN launches of one DLA-34 frame under the tuning knobs of
ct_set_tuning (tile shape, pinned prefetch, split-K target).  Times each distinct layer shape
in isolation with HIP events (back-to-back launches on the torch stream).

    python tools/kbench.py [--batch 1] [--size 512] [--reps 30]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

from centertrack_amd import _lib, ops  # noqa: E402


def time_call(fn, reps):
    """us per launch, measured as the replay time of a HIP graph holding `reps` back-to-back launches
    (device time incl. the inter-kernel boundary, without the host launch cost of this Python loop)."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--layers', default='', help='substring filter on the layer name')
    ap.add_argument('--xcd', action='store_true', help='A/B of the XCD-aware workgroup order only')
    ap.add_argument('--no-dcn', action='store_true')
    ap.add_argument('--no-conv', action='store_true')
    ap.add_argument('--dcn-layers', default='', help='substring filter on the DCN layer name')
    ap.add_argument('--dvariant', default='', help='run only this DCN variant')
    ap.add_argument('--off-scale', type=float, default=1.0, help='std (pixels) of the synthetic DCN offsets')
    ap.add_argument('--variant', default='', help='run only this conv variant (e.g. w64x32)')
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda:0')
    N, S = args.batch, args.size
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)

    def tune(**kw):
        for k, v in kw.items():
            _lib.check(lib.ct_set_tuning(k.encode(), v))

    # (name, count per frame, H(in), Cin, Cout, ks, stride)
    convs = [('level0 3x3 16-16', 1, S, 16, 16, 3, 1), ('level1 3x3s2 16-32', 1, S, 16, 32, 3, 2),
             ('l2 3x3s2 32-64', 1, S // 2, 32, 64, 3, 2), ('l2 3x3 64-64', 3, S // 4, 64, 64, 3, 1),
             ('l2 root 1x1 128-64', 1, S // 4, 128, 64, 1, 1), ('l2 proj 1x1 32-64', 1, S // 4, 32, 64, 1, 1),
             ('l3 proj 1x1 64-128', 2, S // 8, 64, 128, 1, 1), ('l3 root 1x1 256-128', 1, S // 8, 256, 128, 1, 1),
             ('l4 proj 1x1 128-256', 2, S // 16, 128, 256, 1, 1), ('l4 root 1x1 512-256', 1, S // 16, 512, 256, 1, 1),
             ('l5 proj 1x1 256-512', 1, S // 32, 256, 512, 1, 1),
             ('l3 3x3s2 64-128', 1, S // 4, 64, 128, 3, 2), ('l3 3x3 128-128', 7, S // 8, 128, 128, 3, 1),
             ('l3 root 1x1 448-128', 1, S // 8, 448, 128, 1, 1),
             ('l4 3x3s2 128-256', 1, S // 8, 128, 256, 3, 2), ('l4 3x3 256-256', 7, S // 16, 256, 256, 3, 1),
             ('l4 root 1x1 896-256', 1, S // 16, 896, 256, 1, 1),
             ('l5 3x3s2 256-512', 1, S // 16, 256, 512, 3, 2), ('l5 3x3 512-512', 3, S // 32, 512, 512, 3, 1),
             ('l5 root 1x1 1280-512', 1, S // 32, 1280, 512, 1, 1),
             ('off 3x3 64-27 @128', 5, S // 4, 64, 27, 3, 1), ('off 3x3 128-27 @64', 6, S // 8, 128, 27, 3, 1),
             ('off 3x3 256-27 @32', 4, S // 16, 256, 27, 3, 1), ('off 3x3 512-27 @16', 1, S // 32, 512, 27, 3, 1),
             ('heads.0 3x3 64-1280', 1, S // 4, 64, 1280, 3, 1), ('head.2 1x1 256-2', 5, S // 4, 256, 2, 1, 1)]
    variants = [('auto', {}), ('old', dict(conv_ks=-2)), ('ks0', dict(conv_ks=0)), ('ks1', dict(conv_ks=1)),
                ('ks2', dict(conv_ks=2)), ('ks3', dict(conv_ks=3)), ('ks4', dict(conv_ks=4)),
                ('old/cfg2', dict(conv_ks=-2, conv_cfg=2)), ('old/cfg4', dict(conv_ks=-2, conv_cfg=4)),
                ('old/cfg3', dict(conv_ks=-2, conv_cfg=3)), ('w64x32', dict(algo=202)), ('w32/k2', dict(algo=205)), ('w32/k4', dict(algo=206)),
                ('w16/k4', dict(algo=207)), ('w32/nb2', dict(algo=208)), ('w32/nb4', dict(algo=209)),
                ('w32/nb5', dict(algo=210)), ('w32/nb8', dict(algo=211))]
    if args.xcd:
        variants = [('auto', {}), ('auto/x0', dict(xcd_remap=0)), ('ks1', dict(conv_ks=1)), ('ks1/x0', dict(conv_ks=1, xcd_remap=0)),
                    ('old/cfg0', dict(conv_ks=-2, conv_cfg=0)), ('cfg0/x0', dict(conv_ks=-2, conv_cfg=0, xcd_remap=0)),
                    ('w64x32', dict(algo=202)), ('w64x32/x0', dict(algo=202, xcd_remap=0)),
                    ('w32/k2', dict(algo=205)), ('w32/k2/x0', dict(algo=205, xcd_remap=0))]
    convs = [] if args.no_conv else [c for c in convs if args.layers in c[0]]
    if args.variant:
        variants = [v for v in variants if v[0] == args.variant]
    print('%-24s %3s %8s |' % ('layer', 'n', 'GFLOP') + ''.join(' %12s' % v[0] for v in variants))
    tot = {v[0]: 0.0 for v in variants}
    for name, cnt, H, Cin, Cout, ks, stride in convs:
        x = ops.new_view(N, H, H, Cin, dev)
        x.buf.normal_()
        wraw = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
        w = ops.pack_weight(wraw)
        Ho = H // stride
        out = ops.new_view(N, Ho, Ho, Cout, dev, ld=(Cout + 3) // 4 * 4)
        gf = 2.0 * ks * ks * Cin * Cout * N * Ho * Ho / 1e9
        line = '%-24s %3d %8.3f |' % (name, cnt, gf)
        for vname, kw in variants:
            tune(conv_cfg=-1, conv_pipe=1, conv_small_tiles=256, splitk_target=512, conv_ks=-1, xcd_remap=1)
            algo = kw.get('algo', 0)
            tune(**{k: v for k, v in kw.items() if k != 'algo'})
            if (kw.get('conv_cfg', -1) in (1, 5) and Cout > 32 * 8) or (algo and (ks != 3 or stride != 1 or Cin % 64 or (algo >= 208 and Cin != 64))):
                line += ' %12s' % '-'
                continue
            ww = ops.pack_winograd(wraw) if algo else None        # (kept alive: the descriptor only holds its address)
            d = ops.make_conv_desc(x, w, Cout, ks, stride, out=out, relu=True, workspace=ws, algo=algo, w_wino=ww)
            try:
                t = time_call(lambda: _lib.check(lib.ct_conv2d(ctypes.byref(d), _lib.stream_ptr())), args.reps)
                line += ' %7.1f/%4.0f' % (t, gf / t * 1e3)      # us / TFLOP/s
                tot[vname] += t * cnt
            except _lib.CTError:
                line += ' %12s' % 'err'
        print(line)
        sys.stdout.flush()
    print('%-24s %3s %8s |' % ('SUM(us per frame)', '', '') + ''.join(' %12.1f' % tot[v[0]] for v in variants))
    tune(conv_cfg=-1, conv_pipe=1, conv_small_tiles=256, splitk_target=512, conv_ks=-1, xcd_remap=1)
    if args.no_dcn:
        return

    dcns = [('dcn 512-256 @16', 1, S // 32, 512, 256), ('dcn 256-256 @32', 1, S // 16, 256, 256),
            ('dcn 256-128 @32', 2, S // 16, 256, 128), ('dcn 128-128 @64', 2, S // 8, 128, 128),
            ('dcn 128-64 @64', 4, S // 8, 128, 64), ('dcn 256-64 @32', 1, S // 16, 256, 64),
            ('dcn 64-64 @128', 5, S // 4, 64, 64)]
    dvars = [('auto', {}), ('64/1', dict(algo=64, split=1)), ('64/2', dict(algo=64, split=2)), ('64/4', dict(algo=64, split=4)),
             ('64/8', dict(algo=64, split=8)), ('32x64/1', dict(algo=3264, split=1)), ('32x64/2', dict(algo=3264, split=2)),
             ('32x64/4', dict(algo=3264, split=4)), ('32x64/8', dict(algo=3264, split=8)),
             ('32x128/1', dict(algo=32128, split=1)), ('32x128/4', dict(algo=32128, split=4)),
             ('F32x64/1', dict(algo=3264, split=1, fuse=1)), ('F32x64/2', dict(algo=3264, split=2, fuse=1)),
             ('F32x64/4', dict(algo=3264, split=4, fuse=1)), ('F32x64/8', dict(algo=3264, split=8, fuse=1)),
             ('4x32x64/1', dict(algo=43264, split=1)), ('4x32x64/2', dict(algo=43264, split=2)), ('4x32x64/4', dict(algo=43264, split=4)),
             ('4x32x128/1', dict(algo=432128, split=1)), ('4x32x128/4', dict(algo=432128, split=4)),
             ('4F32x64/1', dict(algo=43264, split=1, fuse=1)), ('4F32x64/2', dict(algo=43264, split=2, fuse=1)),
             ('128/1', dict(algo=128, split=1)),
             # round 6: the persistent form of the 32-pixel shapes (ct_dcn_desc.algo 5xxxx / 6xxxx)
             ('P32x64/1', dict(algo=53264, split=1)), ('P32x64/2', dict(algo=53264, split=2)), ('P32x128/1', dict(algo=532128, split=1)),
             ('P4x32x64/1', dict(algo=63264, split=1))]
    if args.no_conv and not args.dvariant:       # (the DCN study of round 2: 4- vs 8-wave workgroups)
        dvars = [v for v in dvars if v[0] in ('32x64/1', '32x64/2', '32x64/4', 'F32x64/1', 'F32x64/2', '32x128/1', '32x128/4') or v[0].startswith('4')]
    dcns = [c for c in dcns if args.dcn_layers in c[0]]
    if args.dvariant:
        dvars = [v for v in dvars if v[0] in args.dvariant.split(',')]
    print('%-24s %3s %8s |' % ('layer', 'n', 'GFLOP') + ''.join(' %12s' % v[0] for v in dvars))
    dtot = {v[0]: 0.0 for v in dvars}
    for name, cnt, H, Cin, Cout in dcns:
        x = ops.new_view(N, H, H, Cin, dev)
        x.buf.normal_()
        om = ops.new_view(N, H, H, 32, dev)
        om.buf.normal_()
        om.buf[..., :18] *= args.off_scale
        om.buf[..., 18:].sigmoid_()
        w = ops.pack_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05)
        out = ops.new_view(N, H, H, Cout, dev)
        w_off = ops.pack_weight(torch.randn(27, Cin, 3, 3, device=dev) * 0.01)
        b_off = torch.zeros(27, device=dev)
        gf = 2.0 * 9 * Cin * Cout * N * H * H / 1e9
        line = '%-24s %3d %8.3f |' % (name, cnt, gf)
        for vname, kw in dvars:
            if kw.get('algo', 0) in (32128, 432128, 128, 532128) and Cout < 128:
                line += ' %12s' % '-'
                continue
            upc = 64 if kw.get('algo', 0) == 63264 else 32         # (persistent: an even number of step units per split)
            if kw.get('algo', 0) >= 50000 and (Cin // upc) % (2 * kw.get('split', 1)):
                line += ' %12s' % '-'
                continue
            fz = dict(w_off=w_off, b_off=b_off) if kw.get('fuse') else {}
            d = ops.make_dcn_desc(x, om, w, Cout, None, None, True, out, workspace=ws, split_k=kw.get('split', 0),
                                  algo=kw.get('algo', 0), **fz)
            t = time_call(lambda: _lib.check(lib.ct_dcn_v2(ctypes.byref(d), _lib.stream_ptr())), args.reps)
            line += ' %7.1f/%4.0f' % (t, gf / t * 1e3)
            dtot[vname] += t * cnt
        print(line)
        sys.stdout.flush()
    print('%-24s %3s %8s |' % ('SUM(us per frame)', '', '') + ''.join(' %12.1f' % dtot[v[0]] for v in dvars))
    tune(dcn_bn=0)


if __name__ == '__main__':
    main()
