"""marching cubes' count + scan and list passes alone (no emit), on the bench volume's thresholded mask: HIP-event times over 20
repetitions.  For A/B builds of k_mc_list / k_mc_count (IVX_LIB_PATH): a variant that writes a wrong list cannot reach an emit here."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from invesalius3_amd import _lib as L
from invesalius3_amd.device import DeviceVolume, c64

img = bench.synth_v512()
vol = DeviceVolume(img)
vol.threshold(226, 3071, preserve=False)
p, z0 = vol._surface_params(True, 0, 0, True)
src, plane = vol._mc_setup(p, z0)  # (sizes the scratch; no emit anywhere in this script)
lib = L.lib()
n = ctypes.c_int64(0)
L.check(lib.ivx_dev_mc_count_bits(ctypes.byref(p), plane, vol._mc_scratch.ptr, ctypes.byref(n), vol.stream))
ntri = cap = n.value
for rep in range(22):
    with vol.timer.span("count"):
        L.check(lib.ivx_dev_mc_count_bits_async(ctypes.byref(p), plane, vol._mc_scratch.ptr, vol.stream))
    with vol.timer.span("list"):
        L.check(lib.ivx_dev_mc_list(ctypes.byref(p), vol._mc_scratch.ptr, c64(cap), vol.stream))
vol.sync()
t = vol.timer.collect()
print(os.environ.get("IVX_LIB_PATH", "base"), "triangles", ntri, {k: round(float(np.median(v[2:])), 4) for k, v in t.items()})
