import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import scenarios as S
from centertrack_amd.detector import Detector, default_opt
from centertrack_amd.model import DLASegHIP
from oracle import detector as odet
name = sys.argv[1]
case = [c for c in S.e2e_mode_cases() if c['name'] == name][0]
g = json.load(open('tests/golden/e2e_modes.json'))[name]
cal = json.load(open('tests/golden/e2e_modes_calibration.json')).get(name)
sd = S.e2e_mode_state_dict(case, cal)
opt = default_opt(case['heads'], input_h=case['H'], input_w=case['W'], **case['opt'])
model = DLASegHIP(case['heads']); model.load_state_dict(sd)
det = Detector(opt, model=model)
oopt = odet.default_opt(input_h=case['H'], input_w=case['W'], num_classes=case['heads']['hm'], **case['opt'])
orc = odet.Detector(oopt, sd, case['heads'])
for t, (images, meta) in enumerate(S.e2e_mode_frames(case)):
    res = det.run(images, dict(meta))['results']
    want = orc.run(images, dict(meta))
    ref = g[t]
    gd, od = det.impl.last_dets, orc.last_dets
    n = int((od['scores'][0] >= oopt.out_thresh).sum())
    print('frame', t, 'n', n, 'len', len(res), len(ref), 'topk xs equal', np.array_equal(gd['xs'][0,:n], od['xs'][0,:n]), 'ys', np.array_equal(gd['ys'][0,:n], od['ys'][0,:n]),
          'max dscore', float(np.abs(gd['scores'][0,:n]-od['scores'][0,:n]).max()))
    bad = 0
    for i, (a, b) in enumerate(zip(res, ref)):
        if int(a['tracking_id']) != int(b['tracking_id']) or np.abs(np.asarray(a['ct'], np.float64) - np.asarray(b['ct'])).max() > 0.02 or int(a['age']) != int(b['age']):
            print('  i', i, 'got id', a['tracking_id'], 'age', a['age'], 'act', a['active'], 'ct', np.asarray(a['ct']), 'score', float(a['score']),
                  '| ref id', b['tracking_id'], 'age', b['age'], 'act', b['active'], 'ct', b['ct'], 'score', b['score'])
            bad += 1
            if bad > 6: break
    if bad: break
