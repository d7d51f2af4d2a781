"""Device times of the remaining kernels at bench size (resident buffers, HIP events): the view-matrix resampler in its
four interpolation modes, the 3-D mask editing kernels, count_regions, mask area, convolve_non_zero.
python tools/bench_edit.py [n]"""
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
    vol = DeviceVolume(img, spacing=(0.5, 0.5, 0.5))
    vol.threshold(*BONE)
    N = img.size
    lib = L.lib()
    dbl = ctypes.c_double
    out = {"n": n}

    def timed(name, fn, nbytes, reps=5):
        for _ in range(2):
            L.check(fn())
        vol.sync()
        vol.timer.collect()
        for _ in range(reps):
            with vol.timer.span(name):
                L.check(fn())
        vol.sync()
        ms = float(np.median(vol.timer.collect()[name]))
        out[name] = {"ms": round(ms, 4), "algorithmic_GB_s": round(nbytes / ms / 1e6, 1)}

    # view-matrix resampling of the whole volume (a rotation about the centre)
    th = 0.3
    c = (n - 1) * 0.5 * 0.5
    rot = np.array([[np.cos(th), -np.sin(th), 0, 0], [np.sin(th), np.cos(th), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    t1, t2 = np.eye(4), np.eye(4)
    t1[:3, 3], t2[:3, 3] = [c, c, c], [-c, -c, -c]
    m = np.ascontiguousarray(t1 @ rot @ t2)
    sp = (dbl * 3)(0.5, 0.5, 0.5)
    mm = (dbl * 16)(*m.ravel())
    res = DeviceBuffer(N * 2)
    status = DeviceBuffer(64)
    for name, code in (("transform_nearest", 0), ("transform_trilinear", 1), ("transform_tricubic", 2), ("transform_lanczos", 3)):
        timed(name, lambda code=code: lib.ivx_dev_apply_view_matrix_transform(L.I16, vol.image.raw, c64(n), c64(n), c64(n), sp, mm,
                                                                            c64(0), 0, code, dbl(-1024.0), res.ptr, c64(n), c64(n),
                                                                            c64(n), status.ptr, vol.stream), 4 * N, reps=3)
    # mask editing
    filt = DeviceBuffer(1024 * 1024)
    filt.upload((np.random.default_rng(0).random((1024, 1024)) < 0.5).astype(np.uint8))
    view = np.eye(4)
    view[2, 3] = -3.0 * n
    proj = np.array([[1.5, 0, 0, 0], [0, 1.5, 0, 0], [0, 0, -1.0, -2.0], [0, 0, -1.0, 0]])
    wts = np.ascontiguousarray(proj @ view)
    a16, b16 = (dbl * 16)(*wts.ravel()), (dbl * 16)(*view.ravel())
    work = DeviceBuffer(N)

    def cut():
        lib.ivx_memcpy_d2d(work.ptr, vol.mask.raw, ctypes.c_size_t(N), vol.stream)
        return lib.ivx_dev_mask_cut(work.ptr, c64(n), c64(n), c64(n), dbl(0.5), dbl(0.5), dbl(0.5), dbl(1e9), filt.ptr, c64(1024),
                                    c64(1024), a16, b16, 1, vol.stream)

    timed("mask_cut(+copy)", cut, 3 * N)
    timed("mask_copy_only", lambda: lib.ivx_memcpy_d2d(work.ptr, vol.mask.raw, ctypes.c_size_t(N), vol.stream), 2 * N)
    ce = (dbl * 3)(n * 0.25, n * 0.25, n * 0.25)
    timed("brush_r20mm", lambda: lib.ivx_dev_brush_mask(work.ptr, None, c64(n), c64(n), c64(n), sp, ce, dbl(20.0), 1, vol.stream), 80 ** 3)
    pts = np.array([[100.0, 120.0], [900.0, 200.0], [700.0, 950.0], [150.0, 800.0]])
    dp = DeviceBuffer(pts.nbytes)
    dp.upload(pts)
    timed("polygon2mask_1024", lambda: lib.ivx_dev_polygon2mask(c64(1024), c64(1024), dp.ptr, L.ptr(pts), c64(4), filt.ptr, vol.stream), 1024 * 1024)
    lab = DeviceBuffer(N * 4)
    lab.upload((np.random.default_rng(1).random(N) < 0.1).astype(np.int32) * np.random.default_rng(2).integers(1, 500, N, dtype=np.int32))
    cnts, outc, st2 = DeviceBuffer(4096), DeviceBuffer(N * 4), DeviceBuffer(64)
    timed("count_regions_i32", lambda: lib.ivx_dev_count_regions(L.I32, lab.ptr, c64(N), c64(500), cnts.ptr, outc.ptr, st2.ptr, vol.stream), 12 * N)
    K = np.zeros(27)
    K[13], K[4], K[22], K[10], K[16], K[12], K[14] = 1.5, -0.25, -0.25, -0.25, -0.25, -0.25, -0.25
    dk = DeviceBuffer(27 * 8)
    dk.upload(K)
    sb = ctypes.c_size_t(0)
    lib.ivx_mask_area_scratch_bytes(c64(n), c64(n), c64(n), ctypes.byref(sb))
    scr, area = DeviceBuffer(sb.value), DeviceBuffer(64)
    timed("mask_area", lambda: lib.ivx_dev_mask_area_u8(vol.mask.raw, c64(n), c64(n), c64(n), dk.ptr, scr.ptr, area.ptr, vol.stream), N)
    if n <= 512:
        f64 = DeviceBuffer(N * 8)
        f64.upload((vol.download_mask() > 127) * 1.0)
        o64 = DeviceBuffer(N * 8)
        timed("convolve_non_zero_3x3x3", lambda: lib.ivx_dev_convolve_non_zero(f64.ptr, c64(n), c64(n), c64(n), dk.ptr, c64(3), c64(3), c64(3), 1, o64.ptr, vol.stream), 16 * N, reps=3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
