"""pageable host <-> device copies through libivx's lane buffers (csrc/ivx_runtime.hip staged_copy): GB/s for the bench
step's three transfers (268 MB int16 volume up, 134 MB mask and 228 MB triangles down), into fresh and into touched arrays.
    IVX_STAGE_THREADS=8 IVX_STAGE_CHUNK_MB=4 python tools/bench_stage.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from invesalius3_amd import _lib as L
from invesalius3_amd.device import DeviceBuffer

L.require_device()
n_up, n_dn = 512 ** 3 * 2, 512 ** 3 + 6323604 * 36
src = np.random.default_rng(1).integers(0, 255, n_up, dtype=np.uint8)
d = DeviceBuffer(max(n_up, n_dn))
res = {}
for rep in range(4):
    t = time.perf_counter(); d.upload(src); res.setdefault("up", []).append(n_up / (time.perf_counter() - t) / 1e9)
    t = time.perf_counter(); out = d.download((n_dn,), np.uint8); res.setdefault("down_fresh", []).append(n_dn / (time.perf_counter() - t) / 1e9)
    t = time.perf_counter(); d.download((n_dn,), np.uint8, out=out); res.setdefault("down_touched", []).append(n_dn / (time.perf_counter() - t) / 1e9)
    del out
pin = L.pinned_empty((n_up,), np.uint8)
pin[:] = src
t = time.perf_counter(); d.upload(pin); res["up_pinned"] = [n_up / (time.perf_counter() - t) / 1e9]
print("threads", os.environ.get("IVX_STAGE_THREADS", "default"), "chunk", os.environ.get("IVX_STAGE_CHUNK_MB", "4"),
      {k: [round(x, 1) for x in v] for k, v in res.items()})
