#!/usr/bin/env python
"""Where the workgroups of the backbone / heads / stem launches spend their time: builds a debug copy of the library
with -DCT_STAMPS (per-workgroup s_memtime stamps at the phase boundaries of stem_kernel, conv_mfma_kernel,
conv_ksplit_kernel and wino_conv_kernel, ct_common.h) and prints, for every such launch of a plan, the launch's wall
span, how the dispatcher spread its workgroups over XCDs / CUs, and the mean time per phase:
    setup    kernel entry -> first global loads issued (index arithmetic, weight prefetch)
    load     ... -> staged patch visible in LDS (first global round trip + LDS store + barrier)
    loop     ... -> last MFMA issued (all channel chunks)
    xchg     ... -> cross-wave exchange barrier passed (K-split reduction / Winograd row transform)
    epi      ... -> last store issued
        python tools/conv_phases.py --build          (needs hipcc; centertrack_amd/build/dbg/libct_stamps.so)
        python tools/conv_phases.py [--config mot17_512] [--streams 1] [--only wino,conv,stem]     (on the GPU box)"""
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
STAMPED = ['dcn_mfma', 'wino_mfma', 'conv_mfma', 'stem', 'decode']


def build():
    from centertrack_amd import build as b
    b.build()
    os.makedirs(DBG, exist_ok=True)
    procs, objs = [], []
    for stem in STAMPED:
        obj = os.path.join(DBG, stem + '.o')
        objs.append(obj)
        procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc'] + b.FLAGS + ['-x', 'hip', '-DCT_STAMPS', '-c',
                                                                          os.path.join(b.CSRC, stem + '.hip'), '-o', obj]))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit('hipcc failed')
    skip = {s + '.o' for s in STAMPED}
    objs += [o for o in glob.glob(os.path.join(b.PKG, 'build', '*.o')) if os.path.basename(o) not in skip]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', LIB] + objs)
    print(LIB)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--config', default='mot17_512')
    ap.add_argument('--streams', type=int, default=1)
    ap.add_argument('--only', default='stem,conv,wino')
    ap.add_argument('--force', default='', help="launch-name substring=algo[,..]: run those launches with another algo code, e.g. "
                                                 "'level3.tree2=225,level5.t2=113' (3x3 stride-1 launches only keep their weights)")
    ap.add_argument('--reps', type=int, default=1, help='print the mean over this many stamped runs')
    ap.add_argument('--loop', action='store_true', help='also print the loop-internal stamps of the Winograd launches (first two chunks, '
                                                        'wave 0: chunk top -> next patch fetched -> first slab transformed -> last MFMA '
                                                        'issued -> patch stored -> barrier passed)')
    args = ap.parse_args()
    if args.build:
        return build()
    os.environ['CENTERTRACK_LIB'] = LIB
    import numpy as np
    import torch
    import scenarios as S
    from centertrack_amd import weights as W
    from centertrack_amd.model import DLASegHIP
    cfg = S.CONFIGS[args.config]
    heads = S.HEAD_SETS[cfg['heads']]
    model = DLASegHIP(heads)
    model.load_state_dict(W.make_synthetic_state_dict(heads, seed=317))
    model = model.to('cuda')
    B, H, Wd = args.streams, cfg['H'], cfg['W']
    plan = model.get_plan(B, H, Wd, True, True, True)
    x = torch.randn(B, 3, H, Wd, device='cuda')
    model.forward_plan(plan, x, x, torch.zeros(B, 1, H, Wd, device='cuda'))
    torch.cuda.synchronize()
    raw = ctypes.CDLL(LIB)
    forced = [f.split('=') for f in args.force.split(',') if f]
    for l in plan['launches']:
        if l.fn == 'conv':
            for sub, algo in forced:
                if sub in l.name:
                    l.args.algo, l.args.split_k = int(algo), 1
                    l.name = '%s @%s' % (l.name, algo)
    NB, WORDS = 8192, 24
    host = np.zeros(NB * WORDS, dtype=np.uint64)
    kinds = [k for k in args.only.split(',') if k]
    names = ['setup', 'load', 'loop', 'xchg', 'epi']
    print('%-44s %-5s %5s %8s | %7s %7s %7s %7s %7s | %6s %6s | %s' % ('launch', 'kind', 'WGs', 'span us', *names, 'wg us',
                                                                       'late', 'XCDs x CUs used, max WGs on one CU'))
    tot = 0.0
    for l in plan['launches']:
        if l.fn not in ('conv', 'heads', 'stem'):
            continue
        for _ in range(3):
            model._run_plan({'launches': [l]})
        torch.cuda.synchronize()
        for k in ('stem', 'conv', 'wino'):
            assert getattr(raw, 'ct_%s_clear_stamps' % k)() == 0
        model._run_plan({'launches': [l]})
        torch.cuda.synchronize()
        got = None
        for k in ('stem', 'conv', 'wino'):
            assert getattr(raw, 'ct_%s_read_stamps' % k)(host.ctypes.data_as(ctypes.c_void_p), NB) == 0
            st = host.reshape(NB, WORDS).astype(np.int64)
            st = st[st[:, 1] != 0]
            if len(st):
                got = (k, st.copy())
        if got is None or got[0] not in kinds:
            continue
        kind, st = got
        rt0, rt1 = st[:, 0], st[:, 7]
        span = (rt1.max() - rt0.min()) * 10e-3                       # s_memrealtime: 100 MHz
        # s_memtime ticks per us from the two clocks of the same workgroups
        dt_rt = (rt1 - rt0).astype(np.float64) * 10e-3
        dt_mt = (st[:, 6] - st[:, 1]).astype(np.float64)
        tick = float(np.median(dt_mt[dt_rt > 0] / dt_rt[dt_rt > 0])) if (dt_rt > 0).any() else 100.0
        ph = [(st[:, i + 1] - st[:, i]).astype(np.float64) / tick for i in range(1, 6)]
        if kind == 'stem':              # stamps 4 / 5 / 9: after stem 0 / 1 / 2
            ph = [(st[:, 2] - st[:, 1]) / tick, (st[:, 3] - st[:, 2]) / tick, (st[:, 9] - st[:, 3]) / tick,
                  np.zeros(len(st)), (st[:, 6] - st[:, 9]) / tick]
        hw = st[:, 8]
        xcc = (hw >> 32) & 0xf
        cu = (hw >> 8) & 0xf
        sh = (hw >> 12) & 0x1
        se = (hw >> 13) & 0x7
        where = xcc * 1000 + se * 100 + sh * 10 * 2 + cu
        uniq, cnt = np.unique(where, return_counts=True)
        late = np.percentile(rt0 - rt0.min(), 90) * 10e-3
        print('%-44s %-5s %5d %8.1f | %7.2f %7.2f %7.2f %7.2f %7.2f | %6.2f %6.2f | %d x %d, %d' % (
            l.name[:44], kind, len(st), span, *(float(np.mean(p)) for p in ph), float(np.mean(dt_rt)), late,
            len(np.unique(xcc)), len(uniq), int(cnt.max())))
        if args.loop and kind == 'wino' and (st[:, 12] != 0).any():
            for chn in range(2):
                w = st[:, 12 + 6 * chn:18 + 6 * chn].astype(np.float64)
                if (w[:, 0] == 0).all():
                    continue
                d = np.diff(w, axis=1) / tick
                print('      chunk %d: fetch issue %.2f  transform %.2f  MFMAs issued %.2f  store %.2f  barrier %.2f  | chunk %.2f us' % (
                    chn, *(float(np.mean(d[:, i])) for i in range(5)), float(np.mean(w[:, 5] - w[:, 0]) / tick)))
        tot += span
    print('sum of spans %.1f us; s_memtime ticks per us (last launch): %.1f' % (tot, tick))


if __name__ == '__main__':
    main()
