#!/usr/bin/env python
"""Micro-benchmark of ct_decode (graph replay of back-to-back launches) on the BASELINE head-map shapes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

from centertrack_amd import ops  # noqa: E402
from tools.kbench import time_call  # noqa: E402

if __name__ == '__main__':
    dev = torch.device('cuda:0')
    for name, (B, C, h, w) in [('mot b1', (1, 1, 128, 128)), ('mot b8', (8, 1, 128, 128)), ('kitti b4', (4, 3, 96, 320)),
                               ('coco b1', (1, 80, 128, 128)), ('coco b4', (4, 80, 128, 128)), ('nusc b4', (4, 10, 112, 200))]:
        hm = torch.rand((B, C, h, w), device=dev) ** 2
        heads = {'reg': torch.rand((B, 2, h, w), device=dev), 'wh': torch.rand((B, 2, h, w), device=dev),
                 'tracking': torch.rand((B, 2, h, w), device=dev)}
        dec = ops.Decoder(hm, heads, 100)
        t = time_call(dec.run, 20)
        print('%-10s %s: %.1f us per decode' % (name, (B, C, h, w), t))
