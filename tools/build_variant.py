#!/usr/bin/env python
"""Build a VARIANT of libcentertrack_hip.so with extra -D flags on one or more sources, for A/B runs on the GPU box
(`CENTERTRACK_LIB=<path> python bench.py ...`): the variant lives under centertrack_amd/build/variants/ (git-ignored,
travels with the gpurun snapshot).
    python tools/build_variant.py NAME SOURCE.hip -DFLAG[=V] [-DFLAG2 ...] [+ SOURCE2.hip -DFLAG ...]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
from centertrack_amd import build as b  # noqa: E402


def main(name, specs):
    b.build()
    out = os.path.join(b.PKG, 'build', 'variants')
    os.makedirs(out, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    groups, cur = [], []
    for a in specs:
        if a == '+':
            groups.append(cur)
            cur = []
        else:
            cur.append(a)
    groups.append(cur)
    replaced, objs = set(), []
    procs = []
    for g in groups:
        source, flags = g[0], g[1:]
        stem = source.rsplit('.', 1)[0]
        obj = os.path.join(out, '%s_%s.o' % (stem, name))
        extra = ['-x', 'hip'] if source.endswith('.hip') else ['-ffp-contract=off']
        procs.append(subprocess.Popen([hipcc] + b.FLAGS + extra + list(flags) + ['-c', os.path.join(b.CSRC, source), '-o', obj]))
        replaced.add(stem + '.o')
        objs.append(obj)
    for p in procs:
        if p.wait() != 0:
            raise SystemExit('hipcc failed')
    objs += [o for o in glob.glob(os.path.join(b.PKG, 'build', '*.o')) if os.path.basename(o) not in replaced]
    lib = os.path.join(out, 'libcentertrack_hip_%s.so' % name)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', lib] + objs)
    print(lib)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
