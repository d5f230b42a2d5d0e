"""Build recipe: hipcc --offload-arch=gfx950 -> centertrack_amd/libcentertrack_hip.so
(in-tree, so the built library travels with the repo snapshot to the GPU box)."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libcentertrack_hip.so')
SOURCES = ['api.cpp', 'runtime.hip', 'conv_mfma.hip', 'wino_mfma.hip', 'dcn_mfma.hip', 'stem.hip', 'elementwise.hip', 'decode.hip', 'pose.hip', 'host_track.cpp', 'host_preprocess.cpp', 'preprocess.hip', 'flip.hip', 'frame_loop.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
         '-Wno-unused-result']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(PKG, 'build')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, 'ct_common.h'), os.path.join(CSRC, 'ksplit_core.h'),
               os.path.join(ROOT, 'include', 'centertrack_hip.h')]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, src.rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            extra = ['-x', 'hip'] if src.endswith('.hip') else ['-ffp-contract=off']
            cmd = [hipcc] + FLAGS + extra + ['-c', sp, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
