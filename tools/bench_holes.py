"""Mask.fill_holes_auto at bench size: the device pass (resident mask) next to the host step it replaces
(scipy.ndimage.label on the inverted mask, what the reference runs before calling its Rust function)."""
import ctypes
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bench import BONE, synth_v512  # noqa: E402
from invesalius3_amd import _lib as L  # noqa: E402
from invesalius3_amd.device import DeviceVolume, c64  # noqa: E402
from invesalius3_amd.mask import _structure  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    img = synth_v512((n, n, n))
    vol = DeviceVolume(img)
    strct = _structure(3, 1)
    sshape = L.i64(strct.shape)
    out = {"n": n}
    for size in (1000, 10 ** 9):
        ms = []
        for _ in range(4):
            vol.threshold(*BONE)
            vol.sync()
            mod = ctypes.c_int(0)
            t0 = time.perf_counter()
            L.check(L.lib().ivx_dev_fill_holes_auto(vol.mask.ptr, c64(n), c64(n), c64(n), L.ptr(strct), sshape,
                                                    ctypes.c_uint32(min(size, 0xFFFFFFFF)), ctypes.byref(mod), vol.stream))
            vol.sync()
            ms.append((time.perf_counter() - t0) * 1e3)
        out["gpu_ms_size_%d" % size] = round(min(ms), 3)
        out["modified_%d" % size] = bool(mod.value)
    if "--scipy" in sys.argv:
        from scipy import ndimage
        vol.threshold(*BONE)
        m = vol.download_mask()
        t0 = time.perf_counter()
        labels, nl = ndimage.label(~(m > 127), ndimage.generate_binary_structure(3, 1), output=np.uint32)
        out["scipy_label_s"] = round(time.perf_counter() - t0, 2)
        out["nlabels"] = int(nl)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
