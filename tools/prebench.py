"""Device time of ct_preprocess_device (graph replay) and of the host ct_preprocess_image, per frame shape.
usage: python tools/prebench.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centertrack_amd import _lib, autotune                       # noqa: E402
from centertrack_amd.detector import MEAN, STD                   # noqa: E402
from centertrack_amd.image import make_meta                      # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device('cuda:0')
    lut = np.empty((3, 256), np.float32)
    mean, std = np.ascontiguousarray(MEAN.reshape(-1)), np.ascontiguousarray(STD.reshape(-1))
    _lib.check(lib.ct_preprocess_lut(mean.ctypes.data, std.ctypes.data, 3, lut.ctypes.data))
    lut_d = torch.from_numpy(lut).to(dev)
    print('%-28s %10s %10s %10s %12s' % ('frame -> input', 'device us', 'GB/s', 'host us', 'H2D u8 us'))
    for (h, w, ih, iw, flip) in [(1080, 1920, 544, 960, 0), (1080, 1920, 512, 512, 0), (375, 1242, 384, 1280, 1),
                                 (900, 1600, 448, 800, 0), (480, 640, 512, 512, 0)]:
        img = np.random.RandomState(0).randint(0, 256, (h, w, 3)).astype(np.uint8)
        meta = make_meta(ih, iw, h, w)
        t64 = np.ascontiguousarray(meta['trans_input'], np.float64)
        pin = torch.from_numpy(img).pin_memory()
        img_d = pin.to(dev)
        out = torch.empty((2 if flip else 1, 3, ih, iw), device=dev)
        fn = lambda: lib.ct_preprocess_device(img_d.data_ptr(), h, w, w * 3, 3, t64.ctypes.data, iw, ih, lut_d.data_ptr(),
                                              out[0].data_ptr(), out[1].data_ptr() if flip else None, _lib.stream_ptr())
        us = autotune._time_graph(fn, reps=20)
        src = min(h * w, 4 * ih * iw) * 3
        nbytes = src + 4 * 3 * ih * iw * (2 if flip else 1)
        cp = autotune._time_graph(lambda: lib.ct_memcpy_async(img_d.data_ptr(), pin.data_ptr(), h * w * 3, 1,
                                                               _lib.stream_ptr()), reps=10)
        hout = np.empty((2 if flip else 1, 3, ih, iw), np.float32)
        t0 = time.time()
        for _ in range(5):
            lib.ct_preprocess_image(img.ctypes.data_as(ctypes.c_void_p), h, w, w * 3, 3, t64.ctypes.data_as(ctypes.c_void_p),
                                    iw, ih, mean.ctypes.data_as(ctypes.c_void_p), std.ctypes.data_as(ctypes.c_void_p),
                                    hout.ctypes.data_as(ctypes.c_void_p), flip)
        host_us = (time.time() - t0) / 5 * 1e6
        print('%-28s %10.1f %10.0f %10.0f %12.1f' % ('%dx%d -> %dx%d%s' % (w, h, iw, ih, ' +flip' if flip else ''), us,
                                                      nbytes / us / 1e3, host_us, cp))


if __name__ == '__main__':
    main()
