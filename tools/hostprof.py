#!/usr/bin/env python
"""Where does the host time of one StreamDetector.step go?  (perf_counter stamps around the phases)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import scenarios as S  # noqa: E402
from centertrack_amd import weights as W  # noqa: E402
from centertrack_amd import detector as D  # noqa: E402
from centertrack_amd.image import make_meta  # noqa: E402
from centertrack_amd.model import DLASegHIP  # noqa: E402

heads = S.HEAD_SETS['mot']
sd = W.make_synthetic_state_dict(heads, seed=317, hm_gain=11.0)
sd['ltrb_amodal.2.bias'] = torch.tensor([-3.0, -3.0, 3.0, 3.0])
opt = D.default_opt(heads, track_thresh=0.4, pre_thresh=0.5)
model = DLASegHIP(heads)
model.load_state_dict(sd)
det = D.StreamDetector(opt, model=model, num_streams=1)
g = torch.Generator().manual_seed(324)
base = torch.randn((1, 3, 512, 512 + 32), generator=g)
frames = [base[:, :, :, 4 * t:4 * t + 512].contiguous().cuda() for t in range(8)]
meta = [make_meta(512, 512, 1024, 1024)]
for i in range(30):
    det.step(frames[i % 8], meta)
torch.cuda.synchronize()
# monkey-patch timers: wrap the sync call and the native tracker step
import centertrack_amd._lib as L  # noqa: E402
lib = L.load()
acc = {'sync': 0.0, 'track': 0.0, 'prehm': 0.0, 'total': 0.0}
orig_sync = lib.ct_stream_synchronize


def timed_sync(sp):
    t = time.perf_counter()
    r = orig_sync(sp)
    acc['sync'] += time.perf_counter() - t
    return r


class LibProxy(object):
    def __getattr__(self, k):
        if k == 'ct_stream_synchronize':
            return timed_sync
        return getattr(lib, k)


L._lib = LibProxy()
ft = det.fast[0]
o_step, o_pre = ft.step, ft.prehm_params


def t_step(*a, **k):
    t = time.perf_counter()
    r = o_step(*a, **k)
    acc['track'] += time.perf_counter() - t
    return r


def t_pre(*a, **k):
    t = time.perf_counter()
    r = o_pre(*a, **k)
    acc['prehm'] += time.perf_counter() - t
    return r


ft.step, ft.prehm_params = t_step, t_pre
N = 500
t0 = time.perf_counter()
for i in range(N):
    det.step(frames[i % 8], meta)
acc['total'] = time.perf_counter() - t0
for k, v in acc.items():
    print('%-8s %8.1f us / frame' % (k, v / N * 1e6))
print('host work outside sync/track/prehm: %.1f us' % ((acc['total'] - acc['sync'] - acc['track'] - acc['prehm']) / N * 1e6))
