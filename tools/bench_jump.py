"""Time the resident (device-level) jump flooding and the directed flood of floodfill_auto_threshold.
usage: python tools/bench_jump.py [size=256]  -> one JSON line"""
import ctypes
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from invesalius3_amd import _lib as L  # noqa: E402
from invesalius3_amd.device import DeviceBuffer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    L.require_device()
    lib = L.lib()
    shape = (n, n, n)
    rng = np.random.default_rng(0)
    nsites = 64
    sites = np.ascontiguousarray(rng.integers(0, n, (nsites, 3)).astype(np.int32))
    d = DeviceBuffer(n ** 3 * 4)
    o = DeviceBuffer(n ** 3 * 4)
    res = {"size": n, "sites": nsites}
    for normalize in (0, 1):
        ts = []
        for rep in range(4):
            d.upload(np.full(shape, -1.0, np.float32))
            o.upload(np.zeros(shape, np.int32))
            L.synchronize()
            t0 = time.perf_counter()
            L.check(lib.ivx_dev_jump_flooding(d.ptr, o.ptr, L.i64(shape), L.ptr(sites), ctypes.c_int64(nsites), normalize, None))
            L.synchronize()
            ts.append(time.perf_counter() - t0)
        res["jump_flooding_normalize%d_ms" % normalize] = round(min(ts) * 1e3, 3)
    res["passes"] = int(np.floor(np.log2(n)))
    res["gvoxel_pass_per_s"] = round(n ** 3 * res["passes"] / (res["jump_flooding_normalize0_ms"] * 1e-3) / 1e9, 2)
    d.close()
    o.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
