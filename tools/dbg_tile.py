import sys, ctypes, numpy as np
sys.path.insert(0, '/root/repo')
from scipy.ndimage import generate_binary_structure
from invesalius3_amd.device import DeviceVolume
from invesalius3_amd import _lib as L
img = np.full((16, 16, 64), 100, np.int16)
vol = DeviceVolume(img)
for rep in range(3):
    vol.out_mask.zero(vol.stream)
    vol.region_grow([(0, 0, 0)], 50, 150, generate_binary_structure(3, 3), fill=1, select_value=None)
    out = (ctypes.c_uint64 * 16)()
    L.lib().ivx_debug_read(out)
    t = [int(v) for v in out[:5]]
    print("cycles: stage %d  local %d  publish %d  mark %d   total %d" % (t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[4]-t[0]))
