#!/usr/bin/env python
"""One JSON line that characterises the GPU box a measurement ran on: sustained fp32 MFMA rate (tools/micro/mfma_peak.hip,
2 WG/CU, MFMA only), device-to-device copy bandwidth (1 GiB) and the duration of an empty-kernel graph node.  The MI355X
boxes of the pool differ by 5-15 % on memory-bound kernels; profiles/ records this line beside every sweep."""
import ctypes
import json
import os
import subprocess
import sys

import torch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro')
SO = os.path.join(HERE, 'libmfma_peak.so')
if not os.path.exists(SO):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', SO,
                           os.path.join(HERE, 'mfma_peak.hip')])
lib = ctypes.CDLL(SO)
lib.mfma_peak.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(1 << 20, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=1):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


iters = 100000
ms = timed(lambda: lib.mfma_peak(0, 512, iters, out.data_ptr(), st))
mfma = 512 * 4 * iters * 16 * 2.0 * 16 * 16 * 4 / ms / 1e9
a = torch.empty(1 << 28, device='cuda')
b = torch.empty(1 << 28, device='cuda')
ms = timed(lambda: b.copy_(a), reps=5)
d2d = 2.0 * a.numel() * 4 / ms / 1e6          # read + write, GB/s
small = torch.empty(1 << 22, device='cuda')   # 16 MiB: the size of one 128x128x64 activation x 4 (L2 / MALL resident)
small2 = torch.empty(1 << 22, device='cuda')
ms = timed(lambda: small2.copy_(small), reps=50)
d2d_small = 2.0 * small.numel() * 4 / ms / 1e6
name = torch.cuda.get_device_name(0)
print(json.dumps({'box_calibration': {'device': name, 'mfma_f32_tflops': round(mfma, 1), 'd2d_1GiB_GBps': round(d2d, 0),
                                      'd2d_16MiB_GBps': round(d2d_small, 0)}}))
