#!/usr/bin/env python3
"""BASELINE configs[4]: MIP family on a resident 512^3 int16 volume, 3-axis sweep.  Kernel-only times from HIP events on
the launch stream.  (The 2048^2 viewport of the config is 4x4 rays per output pixel of the 512^2 projection -- axis-
aligned rays make the supersampling a nearest-neighbour upscale, so the projection itself is what is timed.)"""
import ctypes
import json
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
sys.path.insert(0, ".")
from bench import synth_v512  # noqa: E402
from invesalius3_amd import _lib as L  # noqa: E402
from invesalius3_amd.device import DeviceBuffer, DeviceVolume, c64  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    img = synth_v512((n, n, n))
    vol = DeviceVolume(img)
    lib = L.lib()
    out = DeviceBuffer(n * n * 8)
    tmp = DeviceBuffer(n * n * n * 2)
    small = DeviceBuffer(256)
    mm = small.at(0)
    status = small.at(64)
    reps = 10
    res = {}

    def timed(name, fn, nbytes):
        for _ in range(2):
            fn()
        vol.sync()
        for _ in range(reps):
            with vol.timer.span(name):
                fn()
        ms = float(np.mean(vol.timer.collect()[name]))
        res[name] = {"ms": round(ms, 4), "Mvoxel/s": round(n ** 3 / ms / 1e3, 1), "GB/s": round(nbytes / ms / 1e6, 1),
                     "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000.0, 4)}

    vb = n ** 3 * 2
    for axis in range(3):
        for opn, op in (("max", L.MIP_MAX), ("mean", L.MIP_MEAN)):
            timed("%sip_axis%d" % (opn, axis),
                  lambda: L.check(lib.ivx_dev_mip_reduce(L.I16, vol.image.ptr, c64(n), c64(n), c64(n), axis, op, out.ptr, vol.stream)), vb)
    timed("minmax_prepass", lambda: L.check(lib.ivx_dev_minmax_f32(L.I16, vol.image.ptr, c64(n ** 3), mm, vol.stream)), vb)
    for axis in range(3):
        timed("mida_axis%d" % axis,
              lambda: L.check(lib.ivx_dev_mida(L.I16, vol.image.ptr, c64(n), c64(n), c64(n), axis, ctypes.c_float(300), ctypes.c_float(600),
                                              mm, L.I16, out.ptr, status, vol.stream)), vb)
        timed("lmip_axis%d" % axis,
              lambda: L.check(lib.ivx_dev_lmip(L.I16, vol.image.ptr, c64(n), c64(n), c64(n), axis, ctypes.c_double(700), ctypes.c_double(3033),
                                              out.ptr, vol.stream)), vb)
    timed("fcm_volume_axis0",
          lambda: L.check(lib.ivx_dev_fcm_volume(L.I16, vol.image.ptr, c64(n), c64(n), c64(n), ctypes.c_float(1.0), 0, tmp.ptr, status, vol.stream)), 2 * vb)
    print(json.dumps({"config": "configs[4]: %d^3 int16, projections along each axis" % n, "results": res}, indent=1))


if __name__ == "__main__":
    main()
