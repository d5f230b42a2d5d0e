#!/usr/bin/env python
"""Compare two device assembly listings (`hipcc ... --cuda-device-only -S`) kernel by kernel: which kernels kept the
exact instruction stream (labels renumbered, comments dropped), which changed, which are new -- the check that adding a
variant left the shipped kernels alone.          python tools/isa_diff.py BASE.s NEW.s"""
import re
import sys


def kernels(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
        body = re.sub(r';[^\n]*', '', m.group(2))
        body = re.sub(r'\.L\w+', 'L', body)
        out[m.group(1)] = [l.strip() for l in body.splitlines() if l.strip()]
    return out


def main(base, new):
    a, b = kernels(base), kernels(new)
    changed = 0
    for k in a:
        if k not in b:
            print('GONE   ', k)
            changed += 1
        elif a[k] != b[k]:
            print('CHANGED', k, len(a[k]), '->', len(b[k]))
            changed += 1
        else:
            print('same   ', k, len(a[k]))
    for k in b:
        if k not in a:
            print('new    ', k, len(b[k]))
    return 1 if changed else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1], sys.argv[2]))
