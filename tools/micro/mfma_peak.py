#!/usr/bin/env python
"""Sustained fp32 MFMA rate of the box in kernel-shaped loops (calibration of what `peak` means for the roofline
fractions): python tools/micro/mfma_peak.py   (builds tools/micro/libmfma_peak.so with hipcc on first use)"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libmfma_peak.so')
if not os.path.exists(SO):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', SO,
                           os.path.join(HERE, 'mfma_peak.hip')])
lib = ctypes.CDLL(SO)
lib.mfma_peak.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(1 << 20, device='cuda')
names = ['mfma only, 2 chains', 'mfma only, 4 chains', '+ ds_read_b128 A frags / 8 mfma', '+ barrier / 16 mfma',
         '+ barrier / 32 mfma', '+ barrier / 32 mfma, 4 chains']
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for wgs_per_cu in (1, 2, 4):
    for v, name in enumerate(names):
        per_iter = 16 if v < 4 else 32
        for iters in (2000, 200000):
            blocks = 256 * wgs_per_cu
            lib.mfma_peak(v, blocks, 100, out.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.mfma_peak(v, blocks, iters, out.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            flops = blocks * 4 * iters * per_iter * 2.0 * 16 * 16 * 4
            print('%d WG/CU  %-34s iters %7d  %8.3f ms  %6.1f TFLOP/s' % (wgs_per_cu, name, iters, ms, flops / ms / 1e9))
    sys.stdout.flush()
