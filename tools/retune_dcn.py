#!/usr/bin/env python
"""Re-measure the DCN schedule knobs (`dcnplan5:N,H,W`, seven knobs; round 6 added offset mode 3 and the persistent MAIN
launches, the latter only searched with CENTERTRACK_DCN_TUNE_PERSIST=1) for every (streams, size) the pinned table holds and write the
merged table: run on the GPU box with CENTERTRACK_TUNE_CACHE=<out.json>; the conv entries of the pinned table are
kept, older `dcnplan3:*` / `dcnplan4:*` / `dcnplan5:*` entries of a re-measured shape dropped.  Optional further arguments: N,H,W shapes to restrict the run to.     python tools/retune_dcn.py gpurun_out/tune_new.json"""
import json
import os
import sys

out = sys.argv[1]
os.environ['CENTERTRACK_TUNE_CACHE'] = out
os.environ['CENTERTRACK_DCN_RETUNE'] = '1'
os.environ.setdefault('CENTERTRACK_TUNE_VERBOSE', '')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import scenarios as S  # noqa: E402
from centertrack_amd import autotune, weights as W  # noqa: E402
from centertrack_amd.model import DLASegHIP  # noqa: E402

table = autotune._read_table(autotune.PINNED_TABLE)
shapes = sorted({tuple(int(v) for v in k.split(':')[1].split(',')) for k in table if k.startswith('dcnplan')},
                key=lambda t: t[0] * t[1] * t[2])
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]]
autotune._load_file()
for k in [k for k in autotune._CACHE if k.startswith('dcnplan')]:      # (measured again below)
    del autotune._CACHE[k]
heads = S.HEAD_SETS['mot']
sd = W.make_synthetic_state_dict(heads, seed=317)
for (N, H, Wd) in shapes:
    model = DLASegHIP(heads)
    model.load_state_dict(sd)
    model = model.to('cuda')
    plan = model.get_plan(N, H, Wd, True, True, True)
    print('dcnplan5:%d,%d,%d -> %s' % (N, H, Wd, (plan['dcn_knobs'],)), flush=True)
    del plan, model
    torch.cuda.empty_cache()
merged = {k: list(v) for k, v in table.items() if not k.startswith('dcnplan')}       # (the cache file holds un-pinned keys only)
merged.update({k: list(v) for k, v in autotune._CACHE.items() if not k.startswith(('dcnplan2:', 'dcnplan3:', 'dcnplan4:'))})
# shapes that were not re-measured keep their older schedules
done = {k.split(':')[1] for k in merged if k.startswith('dcnplan5:')}
merged.update({k: list(v) for k, v in table.items() if k.startswith('dcnplan') and k.split(':')[1] not in done})
with open(out, 'w') as f:
    json.dump(dict(sorted(merged.items())), f, indent=0)
print('%d keys -> %s' % (len(merged), out))
