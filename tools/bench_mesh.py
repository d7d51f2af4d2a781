"""Times the indexed-mesh path (point merge) next to the soup on the bench volume.  python tools/bench_mesh.py [n]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bench import synth_v512  # noqa: E402
from invesalius3_amd.device import DeviceVolume  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    img = synth_v512((n, n, n))
    vol = DeviceVolume(img, spacing=(0.5, 0.5, 0.5))
    vol.threshold(226, 3071)
    out = {}
    for name, fn in (("soup", vol.marching_cubes), ("indexed", vol.marching_cubes_indexed)):
        fn()
        vol.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            r = fn()
        vol.sync()
        out[name] = {"ms": (time.perf_counter() - t0) * 100.0, "result": r}
    nv, nt = out["indexed"]["result"]
    out["bytes_soup"] = nt * 36
    out["bytes_indexed"] = nv * 12 + nt * 12
    verts, faces = vol.marching_cubes_indexed(download=True)
    soup = vol.marching_cubes(download=True)
    out["verts_faces_equal_soup"] = bool(np.array_equal(verts[faces], soup))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
