#!/usr/bin/env python
"""Second stage of the DCN schedule choice for the shapes the pinned table already holds (round 6): the pinned six knobs of
`dcnplan4:N,H,W` timed with one-workgroup-per-tile MAIN launches and with persistent ones (knobs[6] = 1; bit-identical
results, so nothing about the summation order changes), the faster written as `dcnplan5:N,H,W` (seven knobs).
    python tools/tune_persist.py <out.json> [N,H,W ..]        (GPU box; merge with tools/merge_tune.py <out.json> --only dcnplan5)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import scenarios as S  # noqa: E402
from centertrack_amd import autotune, weights as W  # noqa: E402
from centertrack_amd.model import DLASegHIP  # noqa: E402


def main():
    out = sys.argv[1]
    table = autotune._read_table(autotune.PINNED_TABLE)
    shapes = sorted({tuple(int(v) for v in k.split(':')[1].split(',')) for k in table if k.startswith('dcnplan4:')},
                    key=lambda t: t[0] * t[1] * t[2])
    if len(sys.argv) > 2:
        shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]]
    heads = S.HEAD_SETS['mot']
    sd = W.make_synthetic_state_dict(heads, seed=317)
    res = {}
    for (N, H, Wd) in shapes:
        base = [int(v) for v in table['dcnplan4:%d,%d,%d' % (N, H, Wd)][:6]]
        us = {}
        for pers in (0, 1):
            os.environ['CENTERTRACK_DCN_KNOBS'] = ','.join(str(v) for v in base + [pers])
            model = DLASegHIP(heads)
            model.load_state_dict(sd)
            model = model.to('cuda')
            plan = model.get_plan(N, H, Wd, True, True, True)
            launches = [l for l in plan['launches'] if l.fn == 'dcn_group' or l.name.endswith('.offset')]
            npers = sum(1 for l in launches if l.fn == 'dcn_group' and int(l.args[0][0].algo) >= 50000)
            us[pers] = min(model._time_launches(launches, reps=10) for _ in range(2)) if (pers == 0 or npers) else float('inf')
            del plan, model, launches
            torch.cuda.empty_cache()
        best = 1 if us[1] < us[0] else 0
        res['dcnplan5:%d,%d,%d' % (N, H, Wd)] = base + [best, round(us[best], 1)]
        print('dcnplan5:%d,%d,%d  per-tile %.1f us, persistent %.1f us -> %s' % (N, H, Wd, us[0], us[1], res['dcnplan5:%d,%d,%d' % (N, H, Wd)]), flush=True)
    with open(out, 'w') as f:
        json.dump(dict(sorted(res.items())), f, indent=0)


if __name__ == '__main__':
    main()
