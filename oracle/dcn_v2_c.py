"""ctypes binding of oracle/dcn_v2_ref.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libdcn_oracle.so')
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.dcn_v2_forward_ref.restype = ctypes.c_int
    return _LIB


def dcn_v2_conv(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    """numpy float32 in / out, same contract as oracle.dcn_v2.dcn_v2_conv"""
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
    rc = lib().dcn_v2_forward_ref(
        x.ctypes.data_as(fp), offset.ctypes.data_as(fp), mask.ctypes.data_as(fp),
        weight.ctypes.data_as(fp), bptr, out.ctypes.data_as(fp),
        B, Ci, H, W, Co, kh, kw, stride, padding, dilation)
    assert rc == 0
    return out
