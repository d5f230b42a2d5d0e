#!/usr/bin/env python
"""Print a compressed instruction schedule (M=mfma G=global_load r=ds_read w=ds_write [waitcnt] |B| barrier)
of one kernel from a hipcc -save-temps gfx950 .s file.  usage: isa_sched.py file.s <kernel-substring>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = re.findall(r'^(_Z\w+):', s, re.M)
names = [n for n in names if pat in n]
for n in names[:1]:
    i = s.index('\n' + n + ':')
    j = s.index('s_endpgm', i)
    lines = s[i:j].split('\n')
    cnt = collections.Counter(l.split()[0] for l in lines if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':'))
    print(n, len(lines), cnt.most_common(14))
    seq = []
    for l in lines:
        t = l.strip().split(' ')[0] if l.strip() else ''
        if t.startswith('v_mfma'): seq.append('M')
        elif t.startswith('global_load'): seq.append('G')
        elif t.startswith('global_store'): seq.append('S')
        elif t.startswith('ds_read'): seq.append('r')
        elif t.startswith('ds_write'): seq.append('w')
        elif t.startswith('s_waitcnt'): seq.append('[' + l.strip().split(' ', 1)[1].replace('vmcnt', 'v').replace('lgkmcnt', 'l').replace(' ', '') + ']')
        elif t.startswith('s_barrier'): seq.append('|B|')
        elif t.startswith('s_cbranch'): seq.append('<br>')
        elif l.strip().endswith(':') and l.startswith('.LBB'): seq.append('\n' + l.strip())
    print(''.join(seq))
