#!/usr/bin/env python
"""Where a DCN workgroup's time goes: builds a debug copy of the library with -DCT_STAMPS (s_memtime stamps at the
phase boundaries of dcn_mfma_kernel, per workgroup) and prints, for every MAIN launch of a plan's DCN schedule and every
layer in it, the mean clocks per phase and the launch's wall span.
    python tools/dcn_phases.py --build          (needs hipcc; the debug library lands in centertrack_amd/build/dbg)
    python tools/dcn_phases.py [--batch 1] [--size 512] [--knobs a,b,c]      (on the GPU box)"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
DBG = os.path.join(ROOT, 'centertrack_amd', 'build', 'dbg')
LIB = os.path.join(DBG, 'libct_stamps.so')


def build():
    from centertrack_amd import build as b
    b.build()
    os.makedirs(DBG, exist_ok=True)
    obj = os.path.join(DBG, 'dcn_mfma.o')
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + b.FLAGS + ['-x', 'hip', '-DCT_STAMPS', '-c',
                                                             os.path.join(b.CSRC, 'dcn_mfma.hip'), '-o', obj])
    objs = [o for o in glob.glob(os.path.join(b.PKG, 'build', '*.o')) if not o.endswith('dcn_mfma.o')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [obj])
    print(LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--knobs', default='')
    args = ap.parse_args()
    if args.build:
        return build()
    os.environ['CENTERTRACK_LIB'] = LIB
    if args.knobs:
        os.environ['CENTERTRACK_DCN_KNOBS'] = args.knobs
    import numpy as np
    import torch
    import scenarios as S
    from centertrack_amd import _lib, weights as W
    from centertrack_amd.model import DLASegHIP
    heads = S.HEAD_SETS['mot']
    model = DLASegHIP(heads)
    model.load_state_dict(W.make_synthetic_state_dict(heads, seed=317))
    model = model.to('cuda')
    plan = model.get_plan(args.batch, args.size, args.size, True, True, True)
    x = torch.randn(args.batch, 3, args.size, args.size, device='cuda')
    model.forward_plan(plan, x, x, torch.zeros(args.batch, 1, args.size, args.size, device='cuda'))
    torch.cuda.synchronize()
    raw = ctypes.CDLL(LIB)
    print('knobs', plan['dcn_knobs'])
    NB, WORDS = 8192, 24
    host = np.zeros(NB * WORDS, dtype=np.uint64)
    names = ['offset conv', 'table', 'prologue', 'loop', 'epilogue']
    for l in plan['launches']:
        if not (l.fn == 'dcn_group' and (l.args[2] & _lib.CT_DCN_MAIN)):
            continue
        for _ in range(3):
            model._run_plan({'launches': [l]})
        torch.cuda.synchronize()
        assert raw.ct_dcn_clear_stamps() == 0
        model._run_plan({'launches': [l]})
        torch.cuda.synchronize()
        assert raw.ct_dcn_read_stamps(host.ctypes.data_as(ctypes.c_void_p), NB) == 0
        st = host.reshape(NB, WORDS).astype(np.int64)
        st = st[st[:, 1] != 0]
        if int(l.args[0][0].algo) >= 50000:        # persistent launch: its own stamp layout (dcn_persist_kernel)
            span = (st[:, 15].max() - st[:, 0].min()) * 10e-3
            print('%s\n  persistent: %d workgroups, wall span %.1f us; start spread max %.1f us' % (
                l.name, len(st), span, (st[:, 0] - st[:, 0].min()).max() * 10e-3))
            for j in range(l.args[1]):
                d = l.args[0][j]
                s = st[st[:, 18] == j]
                if not len(s):
                    continue
                line = '  layer %d: %3d->%3d @%dx%d split %d: %4d WGs, %.1f tiles each (max %d) of %d steps; clk first table+gathers %.0f' % (
                    j, d.Cin, d.Cout, d.H, d.W, max(1, d.split_k), len(s), s[:, 16].mean(), s[:, 16].max(), s[:, 17].mean(),
                    (s[:, 2] - s[:, 1]).mean())
                prev = s[:, 2]
                for i in range(4):
                    ok = s[:, 16] > i
                    if not ok.any():
                        break
                    a3, a4, a5 = s[ok, 3 + 3 * i], s[ok, 4 + 3 * i], s[ok, 5 + 3 * i]
                    mid = s[ok, 19 + i] if i < 2 else None
                    line += '; tile %d: 2 steps%s + table %.0f, loop %.0f (per step %.0f), stores %.0f' % (
                        i, ' %.0f' % (mid - prev[ok]).mean() if mid is not None else '',
                        (a3 - mid).mean() if mid is not None else (a3 - prev[ok]).mean(), (a4 - a3).mean(),
                        (a4 - a3).mean() / max(s[:, 17].mean() - 2, 1), (a5 - a4).mean())
                    prev = s[:, 5 + 3 * i].copy()
                print(line + '; life %.1f us (max %.1f)' % ((s[:, 15] - s[:, 0]).mean() * 10e-3, (s[:, 15] - s[:, 0]).max() * 10e-3))
            continue
        rt0, rt1 = st[:, 0], st[:, 7]
        span = (rt1.max() - rt0.min()) * 10e-3            # s_memrealtime: 100 MHz
        print('%s\n  %d workgroups, wall span %.1f us; start spread p50 %.1f p90 %.1f max %.1f us' % (
            l.name, len(st), span, *(np.percentile(rt0 - rt0.min(), q) * 10e-3 for q in (50, 90, 100))))
        for j in range(l.args[1]):
            d = l.args[0][j]
            s = st[st[:, 8] == j]
            if not len(s):
                continue
            # first round (workgroups the dispatcher started with the launch) against the later ones: cold against warm
            # instruction cache, everybody in phase against staggered
            early = (s[:, 0] - rt0.min()) * 10e-3 < 3.0
            for tag, sel in (('first round', early), ('later     ', ~early)):
                if sel.any() and (~sel).any():
                    php = np.diff(s[sel][:, 1:7], axis=1).mean(axis=0)
                    print('    %s %4d WGs: clk ' % (tag, sel.sum()) + ', '.join('%s %.0f' % (n, v) for n, v in zip(names, php)))
            ph = np.diff(s[:, 1:7], axis=1)
            tot = s[:, 6] - s[:, 1]
            steps = s[:, 9].mean()
            print('  layer %d: %3d->%3d @%dx%d split %d fuse %d: %4d WGs, %4.0f steps; clk ' % (
                j, d.Cin, d.Cout, d.H, d.W, max(1, d.split_k), int(bool(d.fuse_offset)), len(s), steps)
                + ', '.join('%s %.0f' % (n, v) for n, v in zip(names, ph.mean(axis=0)))
                + '; per step %.0f; total %.0f (max %.0f); life %.1f us (max %.1f)' % (
                    ph[:, 3].mean() / max(steps, 1), tot.mean(), tot.max(),
                    (s[:, 7] - s[:, 0]).mean() * 10e-3, (s[:, 7] - s[:, 0]).max() * 10e-3))


if __name__ == '__main__':
    main()
