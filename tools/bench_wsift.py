"""IFT watershed flood on the GPU at BASELINE configs[2] sizes: stage times (HIP events inside the library) and,
for sizes scipy finishes quickly, the mismatch count against live scipy.  python tools/bench_wsift.py 256 512 [--scipy]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bench import synth_v512  # noqa: E402
from invesalius3_amd import _lib as L, watershed_process as wp  # noqa: E402
from scipy.ndimage import generate_binary_structure  # noqa: E402


def markers_for(img):
    mk = np.zeros(img.shape, np.int8)
    d, h, w = img.shape
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    z, y, x = min(max(z, 2), d - 3), min(max(y, 2), h - 3), min(max(x, 2), w - 3)
    for cz in (0, d - 5):
        for cy in (0, h - 5):
            for cx in (0, w - 5):
                mk[cz:cz + 5, cy:cy + 5, cx:cx + 5] = 2
    mk[z - 2:z + 3, y - 2:y + 3, x - 2:x + 3] = 1
    return mk


def main():
    L.require_device()
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [256]
    conns = [int(a[5:]) for a in sys.argv[1:] if a.startswith("conn=")] or [1, 3]
    for n in sizes:
        img = synth_v512((n, n, n))
        cost = (img - img.min()).astype(np.uint16)
        mk = markers_for(img)
        for conn in conns:
            s = generate_binary_structure(3, conn)
            wp.watershed_ift(cost[:16], mk[:16], s)  # warm up
            t = time.perf_counter()
            got, st = wp.watershed_ift(cost, mk, s, want_stats=True)
            wall = time.perf_counter() - t
            rec = dict(n=n, conn=conn, wall_s=round(wall, 4), label1=int((got == 1).sum()), **st)
            if "--scipy" in sys.argv:
                from scipy import ndimage
                t = time.perf_counter()
                sci = ndimage.watershed_ift(cost, mk, s)
                rec["scipy_s"] = round(time.perf_counter() - t, 3)
                rec["mismatch_vs_scipy"] = int((sci != got).sum())
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
