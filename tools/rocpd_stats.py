#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table
(name, calls, total/avg/min/max duration, % of GPU time), like `--stats` CSV output."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\b(void|at::native::|at::)\b', '', name).strip()
    return name[:110]


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    print('%-112s %8s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-112s %8d %12.1f %10.2f %10.2f %10.2f %6.2f' % (short(n), a[0], a[1] / 1e3, a[1] / a[0] / 1e3,
                                                               a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
    print('TOTAL kernel time %.1f us over %d dispatches' % (tot / 1e3, len(rows)))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
