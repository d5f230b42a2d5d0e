import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import scenarios as S
from _parity import calibrated_state_dict, scrolled_stream
from centertrack_amd.detector import StreamDetector, default_opt
from centertrack_amd.image import make_meta
from centertrack_amd.model import DLASegHIP
cfg = S.CONFIGS['mot17_512']; heads = S.HEAD_SETS['mot']
sd = calibrated_state_dict('mot17_512', heads)
for B in (1, 3):
  for graph in (True, False):
    opt = default_opt(heads, track_thresh=0.4, pre_thresh=0.5, zero_tracking=True)
    model = DLASegHIP(heads); model.load_state_dict(sd)
    det = StreamDetector(opt, model=model, num_streams=B, use_graph=graph)
    meta = make_meta(512, 512, 1024, 1024)
    fr = [scrolled_stream(512, 512, 4, 400 + 10 * s) for s in range(B)]
    for t in range(4):
        x = torch.cat([fr[s][t] for s in range(B)], 0)
        det.step(x, [dict(meta) for _ in range(B)])
        d = det.last_dets
        torch.cuda.synchronize()
        tm = det._ctx['merged']['tracking']
        print('B', B, 'graph', graph, 'frame', t, 'rows max|tracking|', float(np.abs(d['tracking']).max()), 'map max', float(tm.abs().max()), 'native', det.native)
