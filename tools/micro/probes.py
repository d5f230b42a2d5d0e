#!/usr/bin/env python
"""Build and load tools/micro/libct_probes.so: the box probes (`ctp_*`) that tools/box_calib.py runs.  They are
diagnostics of the box a measurement ran on, not part of the product ABI (include/centertrack_hip.h); the library is
built in-tree by `python tools/micro/probes.py` (and by __graft_entry__.build()), so it travels with the gpurun snapshot."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'probes.hip')
LIB = os.path.join(HERE, 'libct_probes.so')
_lib = None


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-x', 'hip', SRC, '-o', LIB])
    return LIB


def load():
    """the probe library (built on first use when hipcc is there); raises OSError when it can be neither found nor built"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        try:
            build()
        except (OSError, subprocess.CalledProcessError) as e:
            if not os.path.exists(LIB):
                raise OSError('tools/micro/libct_probes.so is missing and could not be built: %r' % (e,))
    lib = ctypes.CDLL(LIB)
    i, p, sz, u = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint
    lib.ctp_last_error.restype = ctypes.c_char_p
    lib.ctp_mfma.argtypes = [i, i, p, p]
    lib.ctp_chase.argtypes = [p, i, u, p, p]
    lib.ctp_chase_many.argtypes = [p, i, u, i, p, p]
    lib.ctp_stream.argtypes = [p, p, sz, i, i, p]
    lib.ctp_write.argtypes = [p, sz, i, i, i, p, p]
    lib.ctp_ifetch.argtypes = [i, p, p]
    lib.ctp_launches.argtypes = [i, i, p, p]
    lib.ctp_cu_map.argtypes = [i, i, i, p, p]
    lib.ctp_xcd_stream.argtypes = [p, p, sz, i, p, p]
    for name in ('ctp_mfma', 'ctp_chase', 'ctp_chase_many', 'ctp_stream', 'ctp_write', 'ctp_ifetch', 'ctp_launches', 'ctp_cu_map',
                 'ctp_xcd_stream'):
        getattr(lib, name).restype = i
    _lib = lib
    return lib


if __name__ == '__main__':
    print(build(force=True))
