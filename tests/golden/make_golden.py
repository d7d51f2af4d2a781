"""Generates tests/golden/*.npz from the third-party functions the reference itself calls, run in this container
(scipy 1.15.3, numpy 2.2.6; the reference pins scipy 1.14.0): small seeded input / output pairs that travel with the
repository, so the parity tests do not depend on the scipy that happens to be installed where they run.

    python tests/golden/make_golden.py

* watershed_ift.npz  scipy.ndimage.watershed_ift(cost, markers, structure)  (invesalius/data/watershed_process.py:44-46,
                     54-57; the reference fixture of tests/test_segmentation_tools.py:170-213 is case 0), together with the
                     event counts of scipy's unlink defect on each case (oracle.watershed_ift_events): cases whose counts of
                     late / lost pops are zero are the ones every implementation of the documented algorithm must reproduce
* zoom_order2.npz    scipy.ndimage.zoom(a, factor, a.dtype, order=2)  (invesalius/data/imagedata_utils.py:121-130)
"""
import os
import sys

import numpy as np
from scipy import ndimage
from scipy.ndimage import generate_binary_structure

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def ws_cases():
    image = np.zeros((5, 5, 5), dtype=np.int16)
    image[1:4, 1:4, 1:4] = 100
    markers = np.zeros((5, 5, 5), dtype=np.int16)
    markers[2, 2, 2] = 1
    markers[0, 0, 0] = 2
    yield (image - image.min()).astype(np.uint16), markers, generate_binary_structure(3, 1)
    rng = np.random.default_rng(20260924)
    for k in range(11):
        nd = 3 if k % 3 else 2
        shape = tuple(int(v) for v in (rng.integers(3, 12, 3) if nd == 3 else rng.integers(5, 30, 2)))
        hi = int(rng.choice([3, 12, 300, 3000]))
        img = rng.integers(0, hi, shape).astype(np.uint16 if k % 2 else np.uint8)
        if k % 4 == 0:
            img[rng.random(shape) < 0.4] = 0
        mk = np.zeros(shape, np.int16 if k % 2 else np.int8)
        idx = rng.integers(0, img.size, 5)
        mk.ravel()[idx] = rng.choice(np.array([1, 2], mk.dtype), 5)
        yield img, mk, generate_binary_structure(nd, int(rng.integers(1, nd + 1)))


def main():
    from oracle import oracle as orc
    out = {}
    n = 0
    for img, mk, s in ws_cases():
        lab = ndimage.watershed_ift(img, mk, s)
        _, ev = orc.watershed_ift_events(img, mk, s)
        out["img%d" % n], out["mk%d" % n], out["s%d" % n], out["lab%d" % n] = img, mk, s.astype(np.uint8), lab
        out["ev%d" % n] = np.array(ev, np.int64)
        n += 1
    out["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "watershed_ift.npz"), **out)
    rng = np.random.default_rng(7)
    out, n = {}, 0
    for shape, factor in (((9, 11, 13), 0.5), ((12, 20, 17), 1 / 3.0), ((16, 16, 16), 0.5), ((1, 10, 10), 0.5)):
        a = rng.integers(-1024, 3071, shape).astype(np.int16)
        m = np.where(a > 500, 255, 0).astype(np.uint8)
        for arr in (a, m):
            out["a%d" % n], out["f%d" % n] = arr, np.array(factor)
            out["z%d" % n] = ndimage.zoom(arr, factor, arr.dtype, order=2)
            n += 1
    out["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "zoom_order2.npz"), **out)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
