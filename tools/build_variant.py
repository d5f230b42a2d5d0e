#!/usr/bin/env python
"""Build a VARIANT of libcentertrack_hip.so with extra -D flags on one source, for A/B runs on the GPU box
(`CENTERTRACK_LIB=<path> python bench.py ...`): the variant lives under centertrack_amd/build/variants/ (git-ignored,
travels with the gpurun snapshot).      python tools/build_variant.py NAME SOURCE.hip -DFLAG[=V] [-DFLAG2 ...]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
from centertrack_amd import build as b  # noqa: E402


def main(name, source, flags):
    b.build()
    out = os.path.join(b.PKG, 'build', 'variants')
    os.makedirs(out, exist_ok=True)
    stem = source.rsplit('.', 1)[0]
    obj = os.path.join(out, '%s_%s.o' % (stem, name))
    extra = ['-x', 'hip'] if source.endswith('.hip') else ['-ffp-contract=off']
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + b.FLAGS + extra + list(flags) +
                          ['-c', os.path.join(b.CSRC, source), '-o', obj])
    objs = [o for o in glob.glob(os.path.join(b.PKG, 'build', '*.o')) if os.path.basename(o) != stem + '.o']
    lib = os.path.join(out, 'libcentertrack_hip_%s.so' % name)
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + [obj])
    print(lib)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
