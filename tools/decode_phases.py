#!/usr/bin/env python
"""Where decode stage 2 (one workgroup per image: top-K select, sort, head gathers, packed rows to HBM and to pinned host
memory, end-of-frame flag) spends its time: s_memtime stamps of the debug library (`python tools/conv_phases.py --build`,
-DCT_STAMPS) at its phase boundaries, MOT head set, with and without the host rows.
    python tools/decode_phases.py [--streams 1]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'centertrack_amd', 'build', 'dbg', 'libct_stamps.so')
WORDS = 24
NAMES = ['entry -> keys requested', 'select (histogram)', 'rank the K winners', 'gathers + row stores issued',
         'system fence + barrier', 'counter / flag']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--streams', type=int, default=1)
    args = ap.parse_args()
    os.environ['CENTERTRACK_LIB'] = LIB
    import numpy as np
    import torch
    from centertrack_amd import ops
    raw = ctypes.CDLL(LIB)
    dev = torch.device('cuda:0')
    B, h, w, K = args.streams, 128, 128, 100
    torch.manual_seed(3)
    hm = torch.sigmoid(torch.randn((B, 1, h, w), device=dev) * 1.5 - 3.0)
    heads = {'reg': torch.rand((B, 2, h, w), device=dev), 'wh': torch.rand((B, 2, h, w), device=dev),
             'tracking': torch.rand((B, 2, h, w), device=dev)}
    for host in (False, True):
        F = ops.Decoder.row_floats(heads)
        host_out = torch.zeros((B, K, F), dtype=torch.float32).pin_memory() if host else None
        flag = torch.zeros((4,), dtype=torch.int32).pin_memory() if host else None
        dec = ops.Decoder(hm, heads, K, host_out=host_out, done_flag=flag)
        for _ in range(3):
            dec.run()
        torch.cuda.synchronize()
        acc = np.zeros((B, WORDS))
        reps = 20
        for _ in range(reps):
            raw.ct_decode_clear_stamps()
            dec.run()
            torch.cuda.synchronize()
            buf = np.zeros((B, WORDS), dtype=np.uint64)
            raw.ct_decode_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), B)
            acc += buf.astype(np.float64)
        st = acc / reps
        clk = 100.0 * (st[:, 7] - st[:, 1]).mean() / max(1.0, (st[:, 8] - st[:, 0]).mean())     # s_memtime ticks per us
        print('decode stage 2, %d image(s), rows %s: workgroup life %.1f us (s_memrealtime), %.0f ticks per us'
              % (B, 'to HBM + pinned host + flag' if host else 'to HBM only', (st[:, 8] - st[:, 0]).mean() / 100.0, clk))
        for i, nm in enumerate(NAMES):
            a, b = st[:, i + 1], st[:, i + 2]
            if not host and i >= 4:
                continue
            print('   %-30s %6.2f us' % (nm, (b - a).mean() / clk))


if __name__ == '__main__':
    main()
