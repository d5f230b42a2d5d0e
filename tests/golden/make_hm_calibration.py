"""Per-class calibration of the synthetic heat-map head used by the full-size parity tests.

Random-init weights give every class of the `hm` head its own offset (the head's input is post-ReLU, so
w_c . mean(h) acts as a class bias) and the strongest class would supply all K detections -- a poor exercise of the
cross-class top-K at 80 classes -- while a gain large enough to cross the threshold saturates the scores near 1.
This script runs the CPU ORACLE (no GPU, no product code) once per BASELINE configuration on frame 0 of the test
stream, takes every class's own NMS peaks and solves the affine map logit_c = s_c * (raw_c - b0) + t_c that puts the class's
best peak at a target score (0.8 for a single class; spread from 0.8 down to just below the threshold across the
classes of a multi-class head, so that roughly half the classes detect something) and its m-th peak
(m = max(8, 40 / C)) at the configuration's threshold: about 40 detections above the threshold, of mixed classes,
scores well apart.  It stores the per-class weight scale s_c and bias t_c in tests/golden/hm_calibration.json; tests
apply them with
``_parity.calibrated_state_dict`` to BOTH the HIP model and the oracle (same state dict on both sides: this only
chooses the weights, it pins nothing).

    python tests/golden/make_hm_calibration.py
"""
import json
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from _parity import scrolled_stream  # noqa: E402
import scenarios as S  # noqa: E402
from centertrack_amd import weights as Wt  # noqa: E402
from oracle import dla34  # noqa: E402

B0 = -4.6
N_ABOVE = 40
TOP = 0.8


def logit(p):
    return math.log(p / (1 - p))


def main():
    out = {}
    for name, cfg in S.CONFIGS.items():
        heads = S.HEAD_SETS[cfg['heads']]
        H, W = cfg['H'], cfg['W']
        sd = Wt.make_synthetic_state_dict(heads, seed=317, hm_gain=1.0)
        x = scrolled_stream(H, W, 2, 317 + 7)[0]
        with torch.no_grad():
            raw = dla34.forward(x, x, torch.zeros((1, 1, H, W)), sd, heads)[-1]['hm'][0].double()   # [C,h,w] logits
        C = raw.shape[0]
        keep = (F.max_pool2d(raw[None], 3, 1, 1)[0] == raw)
        m = max(8, int(round(N_ABOVE / C)))          # (a slope from fewer than 8 peaks saturates on other frames)
        rs = np.random.RandomState(5)
        thr = cfg['track_thresh']
        if C == 1:
            tops = np.array([TOP])
        elif C < 20:
            tops = np.linspace(TOP, max(thr - 0.1, 0.05), C)[rs.permutation(C)]
        else:                                        # 80 classes: 20 of them detect something, the rest stay below
            tops = np.concatenate((np.linspace(TOP, thr + 0.02, 20), np.full(C - 20, thr * 0.5)))[rs.permutation(C)]
        # flip_test averages the map with its mirrored twin's, which flattens peaks: aim higher
        lift = 1.5 if cfg['flip'] else 0.0
        scale, bias, above = [], [], 0
        if C >= 20:
            # many classes with a handful of peaks each: per-class slopes from 8 peaks do not carry over to other
            # frames.  Standardise each class over the whole map instead (robust) and map the POOLED peaks: the 5th
            # best to TOP, the 60th to the threshold
            mu, sg = raw.mean(dim=(1, 2)), raw.std(dim=(1, 2))
            z = (raw - mu.view(-1, 1, 1)) / sg.view(-1, 1, 1)
            zk = (F.max_pool2d(z[None], 3, 1, 1)[0] == z)
            pooled = torch.sort(z[zk], descending=True)[0]
            G = (logit(TOP) - logit(thr)) / float(pooled[4] - pooled[59])
            P = logit(TOP) - G * float(pooled[4])
            scale = [float(G / sg[c]) for c in range(C)]
            bias = [float(P - scale[c] * (mu[c] - B0)) for c in range(C)]
            above = int((pooled * G + P >= logit(thr)).sum())
            out[name] = {'scale': scale, 'bias': bias, 'peaks_above_thresh_frame0': above}
            print(name, 'classes %d, peaks above the threshold on frame 0: %d (pooled calibration)' % (C, above))
            continue
        for c in range(C):
            peaks = torch.sort(raw[c][keep[c]], descending=True)[0]
            r1, rm = float(peaks[0]), float(peaks[m - 1])
            hi = logit(float(tops[c])) + lift
            # where the class's m-th peak lands: at the threshold, or (80 classes) 4 logits below its best peak
            lo = min(logit(thr) + lift, hi - 1.0) if C < 20 else hi - 4.0
            s_c = (hi - lo) / (r1 - rm)
            t_c = hi - s_c * (r1 - B0)                                # logit = s_c * (raw - b0) + t_c;  raw = w.h + b0
            scale.append(s_c)
            bias.append(t_c)
            above += int(((s_c * (peaks - B0) + t_c) >= logit(cfg['track_thresh'])).sum())
        out[name] = {'scale': scale, 'bias': bias, 'peaks_above_thresh_frame0': above}
        print(name, 'classes %d, peaks above the threshold on frame 0: %d' % (C, above))
    with open(os.path.join(HERE, 'hm_calibration.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
