"""Times the surface stages after marching cubes on the bench volume: indexed mesh (point merge), keep-largest,
mass properties, context-aware smoothing -- device-resident, events on the volume's stream.
python tools/bench_mesh.py [n]   (the CPU side of the comparison lives in tests/test_gpu_smooth.py, where the oracle may
be used)"""
import ctypes
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bench import synth_v512  # noqa: E402
from invesalius3_amd import _lib as L  # noqa: E402
from invesalius3_amd.device import DeviceBuffer, DeviceVolume  # noqa: E402


def timed(vol, fn, reps=5):
    fn()
    vol.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    vol.sync()
    return (time.perf_counter() - t0) * 1e3 / reps, r


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
    img = synth_v512((n, n, n))
    vol = DeviceVolume(img, spacing=(0.5, 0.5, 0.5))
    vol.threshold(226, 3071)
    lib = L.lib()
    out = {"n": n}
    out["soup_ms"], nt = timed(vol, vol.marching_cubes)
    t_img, nt_img = timed(vol, lambda: vol.marching_cubes(from_binary=False, min_value=226, max_value=3071))
    out["soup_image_two_iso_ms"], out["tris_image_two_iso"] = t_img, nt_img  # the "Default" surface: raw image at both thresholds
    out["indexed_ms"], (nv, nt) = timed(vol, vol.marching_cubes_indexed)
    out["verts"], out["tris"] = nv, nt
    c64 = ctypes.c_int64
    ov, of, mass = DeviceBuffer(nv * 12 + 16), DeviceBuffer(nt * 12 + 16), DeviceBuffer(64)
    nrm = DeviceBuffer(nt * 24 + 16)
    n1, n2, nr = c64(0), c64(0), c64(0)

    def keep():
        L.check(lib.ivx_dev_mesh_keep_largest(vol._verts.ptr, c64(nv), vol._faces.ptr, c64(nt), ov.ptr, c64(nv), of.ptr, c64(nt),
                                              ctypes.byref(n1), ctypes.byref(n2), ctypes.byref(nr), vol.stream))
        return n1.value, n2.value, nr.value

    out["keep_largest_ms"], out["largest"] = timed(vol, keep)

    def massp():
        L.check(lib.ivx_dev_mesh_mass_properties(vol._verts.ptr, vol._faces.ptr, c64(nt), mass.ptr, vol.stream))

    out["mass_ms"], _ = timed(vol, massp)
    vol.sync()
    out["volume_area"] = [float(x) for x in mass.download((8,), np.float64)[:2]]

    def normals():
        L.check(lib.ivx_dev_mesh_face_normals(vol._verts.ptr, L.F32, vol._faces.ptr, c64(nt), nrm.ptr, vol.stream))

    out["normals_ms"], _ = timed(vol, normals)
    verts0 = vol._verts.download((nv, 3), np.float32)
    faces0 = vol._faces.download((nt, 3), np.int32)
    work = DeviceBuffer(nv * 12 + 16)

    def smooth(iters):
        def run():
            L.check(lib.ivx_memcpy_d2d(work.ptr, vol._verts.ptr, ctypes.c_size_t(nv * 12), vol.stream))
            L.check(lib.ivx_dev_context_aware_smoothing(work.ptr, L.F32, c64(nv), vol._faces.ptr, c64(nt), nrm.ptr,
                                                        ctypes.c_double(0.7), ctypes.c_double(3.0), ctypes.c_double(0.5),
                                                        ctypes.c_int(iters), None, None, vol.stream))
        return run

    t0, _ = timed(vol, smooth(0), 3)
    t10, _ = timed(vol, smooth(10), 3)
    out["smooth_setup_ms"] = t0  # copy + topology + seeds + weights
    out["smooth_10_steps_ms"] = t10 - t0
    out["smooth_halfstep_us"] = (t10 - t0) / 20 * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
