#!/usr/bin/env python
"""One frame of a `rocprofv3 --kernel-trace --output-format csv` trace as a timeline: start offset, duration and hardware
queue of every kernel -- which launches of a forked frame graph (model.DCN_FORK) really overlap.
    python tools/timeline.py TRACE.csv [--frame -2]"""
import argparse
import csv
import re


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', name)
    return name[:52]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--frame', type=int, default=-2)
    ap.add_argument('--marker', default='decode_stage2', help='kernel that ends a frame')
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ends = [i for i, r in enumerate(rows) if args.marker in r['Kernel_Name']]
    e = ends[args.frame]
    s = ends[args.frame - 1] + 1
    t0 = int(rows[s]['Start_Timestamp'])
    prev_end = {}
    print('%-52s %5s %6s %9s %8s %8s' % ('kernel', 'queue', 'wgs', 'start us', 'us', 'end us'))
    busy_end = t0
    for r in rows[s:e + 1]:
        a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        wgs = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * max(1, int(r.get('Grid_Size_Y', 1) or 1))
        q = r.get('Queue_Id', '?')
        overlap = '  ||' if a < busy_end - 200 else ''
        busy_end = max(busy_end, b)
        print('%-52s %5s %6d %9.1f %8.1f %8.1f%s' % (short(r['Kernel_Name']), q, wgs, (a - t0) / 1e3, (b - a) / 1e3, (b - t0) / 1e3, overlap))
    print('frame span %.1f us, sum of kernel times %.1f us' % ((int(rows[e]['End_Timestamp']) - t0) / 1e3,
                                                             sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[s:e + 1]) / 1e3))


if __name__ == '__main__':
    main()
