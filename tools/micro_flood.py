"""micro-benchmark: single-tile / few-tile floods, to read per-round kernel durations out of a rocprofv3 kernel trace"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from scipy.ndimage import generate_binary_structure
from invesalius3_amd.device import DeviceVolume

for shape in ((16, 16, 64), (16, 16, 128), (32, 32, 64)):
    img = np.full(shape, 100, np.int16)
    vol = DeviceVolume(img)
    for rep in range(3):
        vol.out_mask.zero(vol.stream)
        r = vol.region_grow([(0, 0, 0)], 50, 150, generate_binary_structure(3, 3), fill=1, select_value=None)
    vol.sync()
    print(shape, "rounds", r, "reached", vol.reached_count())
    vol.close()
