"""The contour MaxIP's exact pass (k_fcm_fix) against the number of pixels its bounds leave open: a flat volume (none), the
bench volume, noise.  Run under tools/prof_kt.sh with IVX_FCM_DEBUG=1 for the counts next to the kernel durations."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from invesalius3_amd import _lib as L  # noqa: E402
from invesalius3_amd.device import DeviceBuffer, c64  # noqa: E402
import bench  # noqa: E402

n = 512
lib = L.lib()
vols = {"flat": np.zeros((n, n, n), np.int16), "bench": bench.synth_v512()[0] if isinstance(bench.synth_v512(), tuple) else bench.synth_v512(),
        "noise": np.random.default_rng(1).integers(-1000, 3000, (n, n, n)).astype(np.int16)}
d = DeviceBuffer(n * n * n * 2)
out = DeviceBuffer(n * n * 2 + 64)
status = DeviceBuffer(64)
for name, v in vols.items():
    d.upload(np.ascontiguousarray(v))
    for e in (1.0, 2.0):
        for _ in range(3):
            L.check(lib.ivx_dev_fcm_maxip(L.I16, d.ptr, c64(n), c64(n), c64(n), ctypes.c_float(e), 0, out.ptr, status.ptr, None))
        L.synchronize()
        print(name, e, "done", file=sys.stderr)
