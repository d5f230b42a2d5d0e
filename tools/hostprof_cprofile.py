#!/usr/bin/env python
"""cProfile of StreamDetector.step at one stream (resident frames): which Python lines the ~70 us of host work per
frame outside the stream sync and the native tracker are."""
import cProfile
import os
import pstats
import sys

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
host = [base[:, :, :, 4 * t:4 * t + 512].contiguous().pin_memory() for t in range(8)]
meta = [make_meta(512, 512, 1024, 1024)]
for i in range(30):
    det.step(host[i % 8], meta, prefetch=host[(i + 1) % 8])
torch.cuda.synchronize()
N = 2000
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    det.step(host[i % 8], meta, prefetch=host[(i + 1) % 8])
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime')
st.print_stats(28)
