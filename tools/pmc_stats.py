#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: dispatches, mean duration and mean
counter values per dispatch (+ a few derived ratios).  usage: pmc_stats.py <counter_collection.csv> [top]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\b(void|at::native::|at::)\b', '', name).strip()
    return name[:70]


def main(path, top=30):
    per = collections.OrderedDict()
    seen = set()
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row['Kernel_Name'])
            a = per.setdefault(k, {'n': 0, 'dur': 0.0, 'c': collections.defaultdict(float)})
            did = row['Dispatch_Id']
            if did not in seen:
                seen.add(did)
                a['n'] += 1
                a['dur'] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
            a['c'][row['Counter_Name']] += float(row['Counter_Value'])
    names = sorted({c for a in per.values() for c in a['c']})
    print('%-72s %6s %9s ' % ('kernel', 'calls', 'avg_us') + ' '.join('%16s' % n[-16:] for n in names))
    for k, a in sorted(per.items(), key=lambda kv: -kv[1]['dur'])[:top]:
        print('%-72s %6d %9.2f ' % (k, a['n'], a['dur'] / a['n']) +
              ' '.join('%16.1f' % (a['c'].get(n, 0.0) / a['n']) for n in names))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
