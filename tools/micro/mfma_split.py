#!/usr/bin/env python
"""fp32 contractions emulated on the bf16 matrix cores (VERDICT r4 "next 9": optional, never the headline) -- a MEASURED
starting point, not a product path: accuracy of a K = 576 contraction (3x3 conv over 64 channels) as plain fp32 MFMA, as 3
bf16 terms and as 6 bf16 terms against float64, and the sustained issue rate of the three instruction mixes with and
without the operand split in the loop.      python tools/micro/mfma_split.py"""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libmfma_split.so')
SRC = os.path.join(HERE, 'mfma_split.hip')
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', SO, SRC])
lib = ctypes.CDLL(SO)
vp = ctypes.c_void_p
lib.split_gemm.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
lib.split_rate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
st = vp(torch.cuda.current_stream().cuda_stream)

M, N, K = 256, 256, 576
g = torch.Generator().manual_seed(5)
A = torch.relu(torch.randn((M, K), generator=g)) * 1.5            # post-ReLU activations
B = (torch.rand((K, N), generator=g) * 2 - 1) / 24.0              # U(+-1/sqrt(fan_in)) weights
ref = A.double().numpy() @ B.double().numpy()
scale = float(np.abs(ref).max())
print('K = %d contraction, |C| max %.3f, rms %.3f' % (K, scale, float(np.sqrt((ref ** 2).mean()))))
Ad, Bd = A.cuda(), B.cuda()
for mode, name in ((0, 'fp32 MFMA 16x16x4'), (3, '3 bf16 terms'), (6, '6 bf16 terms')):
    C = torch.zeros((M, N), device='cuda')
    rc = lib.split_gemm(mode, vp(Ad.data_ptr()), vp(Bd.data_ptr()), vp(C.data_ptr()), M, N, K, st)
    torch.cuda.synchronize()
    assert rc == 0
    err = np.abs(C.cpu().double().numpy() - ref)
    print('%-20s max |err| %.3e  (%.2e of max |C|)   rms err %.3e' % (name, err.max(), err.max() / scale, float(np.sqrt((err ** 2).mean()))))
cpu = (A @ B).double().numpy()
print('%-20s max |err| %.3e' % ('torch CPU fp32 mm', float(np.abs(cpu - ref).max())))

out = torch.zeros(1 << 20, device='cuda')
iters = 200000
for wgs in (1, 2, 4):
    blocks = 256 * wgs
    for mode, split, name in ((0, 0, 'fp32 MFMA'), (3, 0, '3 bf16 terms, operands pre-split'), (3, 1, '3 bf16 terms, A split in the loop'),
                              (6, 0, '6 bf16 terms, operands pre-split'), (6, 1, '6 bf16 terms, A split in the loop')):
        lib.split_rate(mode, split, blocks, 100, vp(out.data_ptr()), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.split_rate(mode, split, blocks, iters, vp(out.data_ptr()), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        flops = blocks * 4 * iters * 2 * 2.0 * 16 * 16 * 32          # fp32-EQUIVALENT flops: two 16 x 16 x 32 slices per iteration and wave
        print('%d WG/CU  %-36s %8.2f ms  %7.1f TFLOP/s fp32-equivalent' % (wgs, name, ms, flops / ms / 1e9))
