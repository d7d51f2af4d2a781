"""Watershed pre/post-processing kernels (everything of do_watershed except the third-party marker flood) and the
mask-level kernels at bench size, device-resident, HIP events.  python tools/bench_ws.py [n]"""
import ctypes
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from bench import BONE, synth_v512  # noqa: E402
from invesalius3_amd import _lib as L  # noqa: E402
from invesalius3_amd.device import DeviceBuffer, DeviceVolume, c64  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    img = synth_v512((n, n, n))
    vol = DeviceVolume(img)
    vol.threshold(*BONE)
    N = img.size
    lib = L.lib()
    cost, grad, tmp = DeviceBuffer(N * 2), DeviceBuffer(N * 2), DeviceBuffer(N)
    tmp.upload(np.random.default_rng(0).integers(0, 3, N, dtype=np.uint8))
    m2, mo, acc = DeviceBuffer(N), DeviceBuffer(N), DeviceBuffer(64)
    m2.upload(np.random.default_rng(1).integers(0, 256, N, dtype=np.uint8))
    size = (ctypes.c_int * 3)(3, 3, 3)
    stages = {
        "lut_u16": (lambda: lib.ivx_dev_lut_u16(vol.image.raw, c64(N), ctypes.c_double(2000.0), ctypes.c_double(300.0), 1, cost.ptr, vol.stream), 4 * N),
        "shift_min_u16": (lambda: lib.ivx_dev_shift_min_u16(vol.image.raw, c64(N), -1024, cost.ptr, vol.stream), 4 * N),
        "morph_gradient_3x3x3": (lambda: lib.ivx_dev_morph_gradient_u16(cost.ptr, c64(n), c64(n), c64(n), size, grad.ptr, vol.stream), 4 * N),
        "watershed_merge": (lambda: lib.ivx_dev_watershed_merge(vol.mask.raw, tmp.ptr, c64(N), 0, vol.stream), 3 * N),
        "mask_boolean_union": (lambda: lib.ivx_dev_mask_boolean(1, vol.mask.raw, m2.ptr, mo.ptr, c64(N), vol.stream), 3 * N),
        "masked_density": (lambda: lib.ivx_dev_masked_density_i16(vol.image.raw, vol.mask.raw, c64(N), acc.ptr, vol.stream), 3 * N),
    }
    out = {"n": n}
    for name, (fn, nbytes) in stages.items():
        for _ in range(2):
            L.check(fn())
        vol.sync()
        vol.timer.collect()
        for _ in range(10):
            with vol.timer.span(name):
                L.check(fn())
        vol.sync()
        ms = float(np.median(vol.timer.collect()[name]))
        out[name] = {"ms": round(ms, 4), "algorithmic_GB_s": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
