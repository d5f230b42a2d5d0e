#!/usr/bin/env python
"""Whole-round arithmetic of the launches of one frame, from a `rocprofv3 --kernel-trace` CSV (output format csv:
*_kernel_trace.csv carries grid, workgroup size, LDS and register counts of every dispatch): workgroups, workgroups
resident per CU by registers and by LDS, rounds = workgroups / (256 x per CU), duration -- which launches run a full round
plus a fraction.        python tools/rounds.py TRACE.csv [--frame -2] [--marker decode_stage2]

Two properties of the tracer's columns on gfx950 (rocprofv3 of ROCm 7.2), calibrated against the code-object metadata (round 6):
`VGPR_Count` is HALF the allocated registers of a wave (stem_kernel<16>: 249 in the code object, 128 in the trace; the multi-chunk
Winograd shape 192 / 96) -- doubled here; `LDS_Block_Size` holds the static LDS only, so kernels with dynamic LDS (the Winograd shapes:
27.6 KB per patch buffer, two of them in the multi-chunk shapes; the DCN kernels) are limited by LDS more than this table says: their
`per CU` is an upper bound."""
import argparse
import csv
import re

CUS, SIMDS, VGPRS, LDS_CU = 256, 4, 512, 160 * 1024


def per_cu(vgpr, agpr, lds, threads):
    waves = (threads + 63) // 64
    alloc = max(8, (vgpr + agpr + 7) // 8 * 8)
    per_simd = min(8, VGPRS // alloc)                    # waves per SIMD by registers
    wg_regs = (per_simd * SIMDS) // waves if waves <= per_simd * SIMDS else 0
    wg_lds = LDS_CU // lds if lds > 0 else 99
    wg_waves = (8 * SIMDS) // waves
    return max(0, min(wg_regs, wg_lds, wg_waves)), per_simd


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', name)
    return name[:58]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--frame', type=int, default=-2, help='which frame of the trace (index into the marker occurrences)')
    ap.add_argument('--marker', default='decode_stage2', help='kernel that ends a frame')
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ends = [i for i, r in enumerate(rows) if args.marker in r['Kernel_Name']]
    e = ends[args.frame]
    s = ends[args.frame - 1] + 1
    print('%-58s %6s %5s %6s %6s %7s %7s' % ('kernel', 'wgs', 'VGPR', 'LDS', 'per CU', 'rounds', 'us'))
    tot = 0.0
    for r in rows[s:e + 1]:
        thr = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
        wgs = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(thr, 1)
        v, a, lds = 2 * int(r['VGPR_Count']), 2 * int(r['Accum_VGPR_Count']), int(r['LDS_Block_Size'])
        pc, _ = per_cu(v, a, lds, thr)
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0
        tot += us
        rounds = wgs / float(CUS * pc) if pc else float('nan')
        print('%-58s %6d %5d %6d %6d %7.2f %7.1f' % (short(r['Kernel_Name']), wgs, v + a, lds, pc, rounds, us))
    print('%d launches, %.1f us of kernel time; frame span %.1f us' %
          (e + 1 - s, tot, (int(rows[e]['End_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1000.0))


if __name__ == '__main__':
    main()
