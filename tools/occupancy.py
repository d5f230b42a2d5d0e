#!/usr/bin/env python
"""Registers, spills, static LDS and the waves-per-SIMD they allow for every kernel of a source, from the device listing
(no GPU needed):  python tools/occupancy.py dcn_mfma.hip [-DFLAG ...]
A 256-thread workgroup puts one wave on each SIMD, so waves per SIMD = workgroups per CU as far as registers go
(512 VGPRs per SIMD lane, allocated in blocks of 8); dynamic LDS is not in the listing.  Why it matters at one stream:
a launch of N workgroups runs in ceil(N / (256 * per_CU)) rounds, and 1024 workgroups at 3 per CU are a full round plus
a third of one."""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
from centertrack_amd import build as b  # noqa: E402


def listing(source, flags=()):
    out = os.path.join(b.PKG, 'build', 'dbg')
    os.makedirs(out, exist_ok=True)
    tag = ''.join(re.sub(r'\W', '_', f) for f in flags)
    dst = os.path.join(out, source.rsplit('.', 1)[0] + (tag and '_' + tag) + '.s')
    cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + b.FLAGS + ['-x', 'hip', '--cuda-device-only', '-S'] + list(flags) + \
          [os.path.join(b.CSRC, source), '-o', dst]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return dst


def kernels(path):
    txt = open(path).read()
    rows = []
    for m in re.finditer(r'- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)', txt, re.S):
        blk = m.group(0)

        def f(key):
            mm = re.search(r'\.%s:\s+(\S+)' % key, blk)
            return mm.group(1) if mm else '0'
        rows.append(dict(name=f('name'), vgpr=int(f('vgpr_count')), spill=int(f('vgpr_spill_count')),
                         lds=int(f('group_segment_fixed_size')), scratch=int(f('private_segment_fixed_size'))))
    return rows


def demangle(names):
    try:
        out = subprocess.check_output(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + names).decode().splitlines()
        return [o.replace('(anonymous namespace)::', '') for o in out]
    except Exception:
        return names


def main(argv):
    source, flags = argv[0], argv[1:]
    rows = kernels(listing(source, flags))
    names = demangle([r['name'] for r in rows])
    print('%-78s %5s %6s %8s %8s %6s' % ('kernel', 'VGPR', 'spill', 'scratch', 'LDS', 'waves'))
    for r, n in zip(rows, names):
        alloc = (r['vgpr'] + 7) // 8 * 8
        waves = min(8, 512 // max(alloc, 8))
        print('%-78s %5d %6d %8d %8d %6d' % (n[:78], r['vgpr'], r['spill'], r['scratch'], r['lds'], waves))


if __name__ == '__main__':
    main(sys.argv[1:])
