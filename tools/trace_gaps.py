#!/usr/bin/env python
"""GPU idle time between consecutive frames from a rocprofv3 rocpd database (--kernel-trace --hip-trace
--memory-copy-trace of bench.py): per frame, the gap between the last kernel of frame t and the first kernel of frame
t+1, the memory copies and the host API calls (hipGraphLaunch, hipStreamSynchronize, hipMemcpyAsync) inside it.
usage: trace_gaps.py <results.db> [frames]"""
import sqlite3
import sys


def cols(cur, t):
    return [r[1] for r in cur.execute('pragma table_info(%s)' % t)]


def main(path, nfr=6):
    con = sqlite3.connect(path)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    views = [n for n in names if not n.startswith('rocpd_') and not n.startswith('sqlite')]
    print('views:', views)
    kc = cols(cur, 'kernels')
    ncol = 'name' if 'name' in kc else 'kernel_name'
    ks = cur.execute('select %s, start, end from kernels order by start' % ncol).fetchall()
    first = [i for i, k in enumerate(ks) if 'render_pre_hm' in k[0]]
    print('%d kernels, %d frames' % (len(ks), len(first)))
    copies = []
    for v in views:
        if 'cop' in v.lower():
            c = cols(cur, v)
            print(v, c)
            try:
                copies = cur.execute('select name, start, end from %s order by start' % v).fetchall()
            except sqlite3.Error as e:
                print('  ', e)
    api = []
    for v in views:
        if v in ('regions', 'regions_and_samples', 'hip_api', 'api'):
            c = cols(cur, v)
            print(v, c)
            try:
                api = cur.execute("select name, start, end from %s where name like 'hip%%' order by start" % v).fetchall()
                break
            except sqlite3.Error as e:
                print('  ', e)
    print('periods (us):', ' '.join('%.0f' % ((ks[b][1] - ks[a][1]) / 1e3) for a, b in zip(first[:-1], first[1:])))
    mid = len(first) // 2
    sel = first[mid - nfr:mid + 1]                        # (the host loop sits in the middle of a bench.py run)
    for a, b in zip(sel[:-1], sel[1:]):
        fr = ks[a:b + 1]                                  # (up to and including the next frame's first kernel)
        period = (ks[b][1] - ks[a][1]) / 1e3
        busy = sum(e - s for _, s, e in fr[:-1]) / 1e3
        gaps = sorted(((fr[i + 1][1] - fr[i][2], i) for i in range(len(fr) - 1)), reverse=True)[:3]
        print('frame: %d kernels, period %.1f us, sum of kernels %.1f us' % (len(fr) - 1, period, busy))
        for g, i in gaps:
            print('  idle %6.1f us between %-34s and %-34s' % (g / 1e3, fr[i][0][:34], fr[i + 1][0][:34]))
            lo, hi = fr[i][2], fr[i + 1][1]
            for n, s, e in copies:
                if lo - 20000 <= s <= hi:
                    print('      copy %-26s start %+7.1f us dur %6.1f us' % (str(n)[:26], (s - lo) / 1e3, (e - s) / 1e3))
            for n, s, e in api:
                if lo - 40000 <= e and s <= hi and any(k in n for k in ('GraphLaunch', 'Synchronize', 'MemcpyAsync', 'EventRecord', 'WaitEvent', 'EventQuery', 'StreamQuery')):
                    print('      api  %-26s start %+7.1f us end %+7.1f us' % (n[:26], (s - lo) / 1e3, (e - lo) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
