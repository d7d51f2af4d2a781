"""Cross-check vectors for the marching-cubes case table: scikit-image's two published tables on small volumes.

    /opt/conda/bin/python3.9 tests/golden/make_golden_mc.py        # scikit-image 0.18.3

The reference contours with VTK (vtkContourFilter; not installed anywhere in this container), so this pins nothing to the
reference.  It relates our generated table (tools/gen_mc_tables.py) to two published ones: `lorensen` (the classic table) and
`lewiner` (topologically consistent).  All marching-cubes variants put their vertices at the same places (linear
interpolation on the grid edges that cross the iso-value); they differ in how the vertices of a cell are joined.  Stored per
case and method: vertices (z, y, x index coordinates) and faces.
"""
import os
import sys

import numpy as np
from scipy import ndimage
from skimage import measure


def cases():
    rng = np.random.default_rng(20260925)
    out = []
    for k in range(4):  # smooth blobs: no ambiguous faces to speak of
        f = ndimage.gaussian_filter(rng.normal(0, 1, (14, 15, 16)), 2.0)
        f = (f - f.min()) / (f.max() - f.min())
        a = (f * 3000 - 1000).astype(np.int16)
        a[[0, -1]] = -1000
        a[:, [0, -1]] = -1000
        a[:, :, [0, -1]] = -1000
        out.append(("smooth%d" % k, a, 226.5))
    for k in range(4):  # binary noise: every ambiguous configuration there is
        a = np.zeros((10, 11, 12), np.uint8)
        a[1:-1, 1:-1, 1:-1] = np.where(rng.random((8, 9, 10)) < (0.3 + 0.1 * k), 255, 0)
        out.append(("binary%d" % k, a, 127.0))
    return out


def main(path):
    data, names = {}, []
    for name, a, iso in cases():
        names.append(name)
        data["vol_" + name] = a
        data["iso_" + name] = np.float64(iso)
        for m in ("lorensen", "lewiner"):
            v, f, _n, _val = measure.marching_cubes(a.astype(np.float32), iso, method=m, allow_degenerate=True)
            data["v_%s_%s" % (m, name)] = v.astype(np.float32)
            data["f_%s_%s" % (m, name)] = f.astype(np.int32)
    data["names"] = np.array(names)
    np.savez_compressed(path, **data)
    print(len(names), "cases")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc_skimage.npz"))
