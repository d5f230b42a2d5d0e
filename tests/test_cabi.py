"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/centertrack_hip.h declares (no compute calls without a GPU); host-side argument
validation returns error codes instead of crashing."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.fixture(scope='module')
def lib():
    from centertrack_amd import build
    path = build.build()
    assert os.path.exists(path)
    return ctypes.CDLL(path)


def _declared():
    src = open(os.path.join(ROOT, 'include', 'centertrack_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ct_[a-z0-9_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), 'missing export ' + n
    from centertrack_amd import _lib
    assert sorted(_lib.EXPORTS) == names, 'python binding list out of sync with the header'


def test_version_and_errors_without_gpu(lib):
    from centertrack_amd import _lib
    l = _lib.load()
    assert l.ct_version() >= 100
    d = _lib.ConvDesc()
    assert l.ct_conv2d(ctypes.byref(d), None) == 1          # CT_ERR_ARG: null pointers
    assert b'null' in l.ct_last_error()
    assert l.ct_packed_weight_elems(27, 64, 3) == 9 * 4 * 2 * 256
    dd = _lib.DecodeDesc()
    assert l.ct_decode(ctypes.byref(dd), None) == 1


def test_product_never_imports_oracle():
    """the product package must not route through the CPU oracle"""
    pkg = os.path.join(ROOT, 'centertrack_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), fn


def test_header_is_plain_c99(tmp_path):
    """the drop-in boundary is a C ABI: include/centertrack_hip.h must compile as C99 (no C++-isms, no torch types)"""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    src = tmp_path / 'hdr.c'
    src.write_text('#include "centertrack_hip.h"\nint main(void) { ct_conv_desc c; ct_dcn_desc d; ct_decode_desc e; '
                   'ct_pose_desc p; (void)c; (void)d; (void)e; (void)p; return ct_version() > 0 ? 0 : 1; }\n')
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I' + os.path.join(root, 'include'),
                        '-fsyntax-only', str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
