#!/usr/bin/env python
"""Per-launch timing of the DCN schedule of one plan (grouped MAIN / FINISH launches, offset convs): name, layers,
GFLOP, us, TFLOP/s.   python tools/dcn_slots.py [--batch 1] [--size 512] [--knobs a,b,c]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import scenarios as S  # noqa: E402
from centertrack_amd import _lib, weights as W  # noqa: E402
from centertrack_amd.model import DLASegHIP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--knobs', default='')
    args = ap.parse_args()
    if args.knobs:
        os.environ['CENTERTRACK_DCN_KNOBS'] = args.knobs
    heads = S.HEAD_SETS['mot']
    model = DLASegHIP(heads)
    model.load_state_dict(W.make_synthetic_state_dict(heads, seed=317))
    model = model.to('cuda')
    plan = model.get_plan(args.batch, args.size, args.size, True, True, True)
    x = torch.randn(args.batch, 3, args.size, args.size, device='cuda')
    model.forward_plan(plan, x, x, torch.zeros(args.batch, 1, args.size, args.size, device='cuda'))
    torch.cuda.synchronize()
    print('knobs', plan['dcn_knobs'])
    tot = 0.0
    for l in plan['launches']:
        if not (l.fn == 'dcn_group' or l.name.endswith('.offset')):
            continue
        us = model._time_launches([l], reps=20)
        gf = 0.0
        wgs = 0
        if l.fn == 'dcn_group' and (l.args[2] & _lib.CT_DCN_MAIN):
            for j in range(l.args[1]):
                d = l.args[0][j]
                hw = d.N * d.H * d.W
                gf += 2e-9 * 9 * d.Cin * d.Cout * hw + (2e-9 * 9 * d.Cin * 27 * hw if d.fuse_offset else 0)
                wgs += d.N * ((d.H + 1) // 2) * ((d.W + 15) // 16) * ((d.Cout + 63) // 64) * max(1, d.split_k)
        tot += us
        print('%-110s %5d WGs %7.3f GFLOP %7.1f us %6.1f TF/s' % (l.name[:110], wgs, gf, us, gf / us * 1e3 if gf else 0))
    print('total %.1f us' % tot)


if __name__ == '__main__':
    main()
