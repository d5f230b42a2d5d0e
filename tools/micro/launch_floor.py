#!/usr/bin/env python
"""What sequencing a chain of small dependent layers costs on this box:
  * a dependent kernel boundary inside a captured HIP graph (N trivial kernels of 1 / 256 / 1024 workgroups, each reading
    what its predecessor wrote): us per launch;
  * a device-wide barrier inside ONE kernel of 256 co-resident workgroups (atomic counter, release / acquire fences at
    agent scope across the 8 XCDs, bounded spin): us per barrier, and whether the neighbour's data always arrived.
    python tools/micro/launch_floor.py      (builds tools/micro/liblaunch_floor.so with hipcc on first use)"""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get('LAUNCH_FLOOR_LIB', os.path.join(HERE, 'liblaunch_floor.so'))
SRC = os.path.join(HERE, 'launch_floor.hip')
if 'LAUNCH_FLOOR_LIB' not in os.environ and (not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC)):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', SO, SRC])


def main():
    lib = ctypes.CDLL(SO)
    vp = ctypes.c_void_p
    lib.launch_tiny.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp]
    lib.launch_barrier.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    buf = torch.zeros(8192, device='cuda')
    ctr = torch.zeros(1, dtype=torch.int32, device='cuda')
    err = torch.zeros(1, dtype=torch.int32, device='cuda')
    N = 200
    import sys
    xcd_only = '--xcd-only' in sys.argv
    for blocks in (() if xcd_only else (1, 256, 1024)):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            lib.launch_tiny(blocks, 4, buf.data_ptr(), vp(side.cuda_stream))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            lib.launch_tiny(blocks, N, buf.data_ptr(), vp(torch.cuda.current_stream().cuda_stream))
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print('graph of %d dependent trivial kernels, %4d workgroups each: %.2f us per launch' % (N, blocks, e0.elapsed_time(e1) * 1e3 / (5 * N)))
    for blocks in (() if xcd_only else (64, 256)):
        for rounds in (10, 1000):
            ctr.zero_()
            err.zero_()
            buf.zero_()
            torch.cuda.synchronize()
            st = vp(torch.cuda.current_stream().cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.launch_barrier(blocks, rounds, buf.data_ptr(), ctr.data_ptr(), err.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            ok = bool((buf[(rounds - 1) % 2 * 4096:(rounds - 1) % 2 * 4096 + blocks] == rounds).all())
            print('one kernel, %3d workgroups, %4d device-wide barriers: %.2f us per barrier (kernel %.1f us); bounded-spin '
                  'timeouts %d, neighbour data %s' % (blocks, rounds, e0.elapsed_time(e1) * 1e3 / rounds, e0.elapsed_time(e1) * 1e3,
                                                      int(err.item()), 'always arrived' if ok else 'STALE'))
    # XCD-local barriers: the workgroups of one XCD meet in their own L2 (workgroup-scope atomics, sc0 loads), all 8 XCDs at once
    lib.launch_barrier_xcd.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    ctr8 = torch.zeros(8 * 32, dtype=torch.int32, device='cuda')
    err2 = torch.zeros(2, dtype=torch.int32, device='cuda')
    for blocks in (64, 256, 512):
        for rounds in (10, 200):
            ctr8.zero_()
            err2.zero_()
            buf.zero_()
            torch.cuda.synchronize()
            st = vp(torch.cuda.current_stream().cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.launch_barrier_xcd(blocks, rounds, buf.data_ptr(), ctr8.data_ptr(), err2.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            ok = bool((buf[(rounds - 1) % 2 * 4096:(rounds - 1) % 2 * 4096 + blocks] == rounds).all())
            print('one kernel, %3d workgroups (%2d per XCD), %4d XCD-LOCAL barriers: %.2f us per barrier (kernel %.1f us); bounded-spin '
                  'timeouts %d, workgroups not on XCD id %% 8: %d, neighbour data %s'
                  % (blocks, blocks // 8, rounds, e0.elapsed_time(e1) * 1e3 / rounds, e0.elapsed_time(e1) * 1e3,
                     int(err2[0].item()), int(err2[1].item()), 'always arrived' if ok else 'STALE'), flush=True)


if __name__ == '__main__':
    main()
