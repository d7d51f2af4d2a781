"""What a union-find restricted to the tiles the coarse pass leaves open could cost at best: the flat,
frontier-free engine (IVX_FLOOD_MODE=ccl, csrc/k_ccl.hip) on the first nz slices of the bench volume -- at nz = 96 the
sub-volume holds about as many words (0.39 M) as the ~1 500 open tiles of the full flood, so its time is what the same
launches cost on that much data, without the extra bookkeeping a tile list would add.  Default engine beside it.
    IVX_FLOOD_MODE=ccl python tools/ccl_size.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.ndimage import generate_binary_structure

import bench
from invesalius3_amd.device import DeviceVolume

img = bench.synth_v512()
S26 = generate_binary_structure(3, 3)
for nz in (48, 96, 192, 512):
    sub = np.ascontiguousarray(img[256 - nz // 2:256 + nz // 2])
    z, y, x = np.unravel_index(int(np.argmax(sub)), sub.shape)
    vol = DeviceVolume(sub)
    for rep in range(8):
        vol.zero_out_mask()
        vol.threshold(226, 3071, preserve=False)
        with vol.timer.span("grow"):
            r = vol.region_grow([(int(x), int(y), int(z))], 226, 3071, S26, fill=1, select_value=254)
    vol.sync()
    t = vol.timer.collect()["grow"]
    print("mode", os.environ.get("IVX_FLOOD_MODE", "rounds"), "nz", nz, "words", sub.size // 64, "rounds", r, "reached", vol.reached_count(),
          "grow ms min/med", round(min(t), 4), round(float(np.median(t)), 4))
    vol.close()
