"""ctypes binding of oracle/_ref/libdcn_v2_ref.so -- the REFERENCE's own compiled DCNv2 CPU forward (upstream
CharlesShang/DCNv2 ``src/cpu``) behind oracle/ref_shim.cpp.  TEST INFRASTRUCTURE ONLY.

The library exists only where ``make -C oracle ref`` found the un-vendored submodule's sources
(``/root/reference/src/lib/model/networks/DCNv2/src/cpu``); they are absent today, so ``available()`` is False and
the test that pins the restatements to upstream is skipped (DCNv2: parity unpinned, see oracle/__init__.py)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, '_ref', 'libdcn_v2_ref.so')
_LIB = None


def available():
    return os.path.exists(PATH)


def lib():
    global _LIB
    if _LIB is None:
        import torch  # noqa: F401  (libtorch must be loaded before the extension build)
        _LIB = ctypes.CDLL(PATH)
        _LIB.dcn_v2_forward_upstream.restype = ctypes.c_int
    return _LIB


def dcn_v2_conv(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    """numpy float32 in / out, same contract as oracle.dcn_v2_c.dcn_v2_conv"""
    x, offset, mask, weight = [np.ascontiguousarray(a, np.float32) for a in (x, offset, mask, weight)]
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    out = np.empty((B, Co, Ho, Wo), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    bptr = None
    if bias is not None:
        bias = np.ascontiguousarray(bias, np.float32)
        bptr = bias.ctypes.data_as(fp)
    rc = lib().dcn_v2_forward_upstream(
        x.ctypes.data_as(fp), offset.ctypes.data_as(fp), mask.ctypes.data_as(fp),
        weight.ctypes.data_as(fp), bptr, out.ctypes.data_as(fp),
        B, Ci, H, W, Co, kh, kw, stride, padding, dilation)
    if rc != 0:
        raise RuntimeError('dcn_v2_forward_upstream failed (%d)' % rc)
    return out
