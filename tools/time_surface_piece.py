#!/usr/bin/env python3
"""where the wall time of the host-level whole-volume surface call goes (512^3 bench mask): the two ivx_marching_cubes calls
(count, then emit) and the result array, timed apart"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bench import BONE, synth_v512  # noqa: E402
from invesalius3_amd import _lib as L, slice_, surface_process as sp  # noqa: E402

n = 512
img = synth_v512((n, n, n))
mask = np.zeros((n + 1,) * 3, np.uint8)
slice_.do_threshold_to_all_slices(mask, img, BONE)
a = mask[1:, 1:, 1:]
p = sp._mc_params(a, (1.0, 1.0, 1.0), [127.0], 0, True, True, True, 0.0, 1)
lib = L.lib()
for rep in range(3):
    cnt = ctypes.c_int64(0)
    t0 = time.perf_counter()
    L.check(lib.ivx_marching_cubes(ctypes.byref(p), L.ptr(a), L.i64(a.strides), None, ctypes.c_int64(0), ctypes.byref(cnt)))
    t1 = time.perf_counter()
    tris = np.empty((cnt.value, 3, 3), np.float32)
    t2 = time.perf_counter()
    m = ctypes.c_int64(0)
    L.check(lib.ivx_marching_cubes(ctypes.byref(p), L.ptr(a), L.i64(a.strides), L.ptr(tris), ctypes.c_int64(cnt.value), ctypes.byref(m)))
    t3 = time.perf_counter()
    print("rep %d: count call %.2f ms, np.empty %.2f ms, emit call %.2f ms (%d triangles, %.0f MB)" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, cnt.value, tris.nbytes / 1e6))
    t4 = time.perf_counter()
    L.check(lib.ivx_marching_cubes(ctypes.byref(p), L.ptr(a), L.i64(a.strides), L.ptr(tris), ctypes.c_int64(cnt.value), ctypes.byref(m)))
    print("       emit call again into the (now touched) array %.2f ms" % ((time.perf_counter() - t4) * 1e3))
