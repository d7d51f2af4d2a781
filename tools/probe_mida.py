"""MIDA over the bench volume (512^3) for windows that end the rays early, late or never: what bounds the walk?  (Device time per
projection from HIP events, best of 5.)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from invesalius3_amd import _lib as L  # noqa: E402
from invesalius3_amd.device import DeviceBuffer, c64  # noqa: E402
import bench  # noqa: E402
import torch  # noqa: E402

n = 512
lib = L.lib()
img = bench.synth_v512()
if isinstance(img, tuple):
    img = img[0]
d = DeviceBuffer(n * n * n * 2)
d.upload(np.ascontiguousarray(img))
out = DeviceBuffer(n * n * 2 + 64)
status = DeviceBuffer(64)
mm = DeviceBuffer(64)
L.check(lib.ivx_dev_minmax_f32(L.I16, d.ptr, c64(n * n * n), mm.ptr, None))
L.synchronize()
for wl, ww in ((300, 300), (300, 400), (40, 400), (-600, 1500), (3000, 10), (32000, 2), (-2000, 2)):
    for axis in (0, 1, 2):
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            L.check(lib.ivx_dev_mida(L.I16, d.ptr, c64(n), c64(n), c64(n), axis, ctypes.c_float(wl), ctypes.c_float(ww), mm.ptr, L.I16, out.ptr,
                                     status.ptr, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("wl %6d ww %5d axis %d: %.4f ms" % (wl, ww, axis, best))
