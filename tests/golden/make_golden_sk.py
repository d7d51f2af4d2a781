"""Golden vectors for the scikit-image branch of do_watershed (watershed_process.py:39,52).

Run with the interpreter that has scikit-image in THIS container (it is not importable from /usr/bin/python3):

    /opt/conda/bin/python3.9 tests/golden/make_golden_sk.py            # scikit-image 0.18.3, numpy 1.26.4

The reference pins scikit-image 0.24.0; what is exercised here is the compiled flood kernel
(`skimage.segmentation._watershed_cy.watershed_raveled`) whose algorithm did not change in between.  Two outputs per case:
  hi_*  skimage.segmentation.watershed(image, markers, bstruct) exactly as the reference calls it;
  lo_*  the same wrapper steps (pad, ravel) around the same compiled kernel, but with the neighbour list in the
        documented order -- raster order of the structure, sorted STABLY by L1 distance.  0.18.3 sorts with numpy's default
        (unstable) argsort, so its order depends on the numpy build: with the numpy 1.26.4 next to it the two orders
        coincide for the 6-neighbour 3-D and 4-neighbour 2-D structures and differ for 8 / 18 / 26 neighbours
        (same_order_* records which).
Nothing here is read at test time except the .npz it writes.
"""
import os
import sys

import numpy as np
import skimage
from scipy.ndimage import generate_binary_structure, morphological_gradient
from skimage.morphology._util import _offsets_to_raveled_neighbors
from skimage.segmentation import _watershed_cy, watershed


def stable_neighbors(padded_shape, selem):
    idx = np.stack(np.nonzero(selem), axis=-1)
    offsets = idx - 1
    factors = np.cumprod((tuple(padded_shape[1:]) + (1,))[::-1])[::-1]
    rav = (offsets * factors).sum(axis=1)
    dist = np.abs(offsets).sum(axis=1)
    rav = rav[np.argsort(dist, kind="stable")]
    return rav[1:].astype(np.intp)


def low_level(image, markers, selem):
    img = np.pad(image.astype(np.float64), 1, mode="constant")
    mask = np.pad(np.ones(image.shape, np.int8), 1, mode="constant").ravel()
    out = np.pad(markers.astype(np.int32), 1, mode="constant")
    nb = stable_neighbors(img.shape, selem)
    strides = np.array(img.strides, dtype=np.intp) // img.itemsize
    _watershed_cy.watershed_raveled(img.ravel(), np.flatnonzero(out).astype(np.intp), nb, mask, strides, 0.0,
                                    out.ravel(), False)
    sl = tuple(slice(1, -1) for _ in image.shape)
    return out[sl].copy(), nb


def cases():
    rng = np.random.default_rng(20260924)
    out = []
    # the reference's own fixture (tests/test_segmentation_tools.py:170-213), through the gradient as do_watershed does
    image = np.zeros((5, 5, 5), np.int16)
    image[1:4, 1:4, 1:4] = 100
    markers = np.zeros((5, 5, 5), np.int16)
    markers[2, 2, 2] = 1
    markers[0, 0, 0] = 2
    grad = morphological_gradient((image - image.min()).astype("uint16"), (3, 3, 3))
    out.append(("ref5", grad, markers, generate_binary_structure(3, 1)))
    k = 0
    for shape in [(6, 7, 8), (9, 10, 11), (12, 12, 12), (5, 16, 16), (16, 17, 3), (1, 9, 9), (3, 3, 3), (20, 21, 22)]:
        for conn in (1, 2, 3):
            for levels in (2, 5, 40, 4000):
                img = rng.integers(0, levels, size=shape).astype(np.uint16)
                if levels == 40:  # gradient of a noise field, as the reference feeds it
                    img = morphological_gradient(img, (3, 3, 3))
                mk = np.zeros(shape, np.int16)
                n_mark = int(rng.integers(2, max(3, img.size // 15)))
                pos = rng.choice(img.size, size=n_mark, replace=False)
                mk.ravel()[pos] = rng.integers(1, 3 if k % 2 else 6, size=n_mark)
                out.append(("r%03d_c%d_l%d" % (k, conn, levels), img, mk, generate_binary_structure(3, conn)))
                k += 1
    # brush-like markers: two blobs of equal-valued voxels (the GUI's case: many tied age-0 entries)
    for conn in (1, 2, 3):
        img = morphological_gradient(rng.integers(0, 6, size=(14, 15, 16)).astype(np.uint16), (3, 3, 3))
        mk = np.zeros(img.shape, np.int16)
        mk[2:5, 2:6, 2:6] = 1
        mk[9:12, 8:13, 9:14] = 2
        out.append(("brush_c%d" % conn, img, mk, generate_binary_structure(3, conn)))
    # 2-D calls (styles.py:1958,1975): a slice and a 3x3 structure
    for conn in (1, 2):
        for levels in (3, 50):
            img = rng.integers(0, levels, size=(17, 19)).astype(np.uint16)
            mk = np.zeros(img.shape, np.int16)
            pos = rng.choice(img.size, size=12, replace=False)
            mk.ravel()[pos] = rng.integers(1, 3, size=12)
            out.append(("s2d_c%d_l%d" % (conn, levels), img, mk, generate_binary_structure(2, conn)))
    return out


def main(path):
    data = {"versions": np.array([skimage.__version__, np.__version__])}
    names = []
    agree = 0
    for name, img, mk, st in cases():
        hi = watershed(img, mk, st)
        lo, nb = low_level(img, mk, st)
        nb_hi = _offsets_to_raveled_neighbors(tuple(s + 2 for s in img.shape), st, center=(1,) * img.ndim)
        agree += int(np.array_equal(hi, lo))
        names.append(name)
        data["img_" + name] = img
        data["mk_" + name] = mk
        data["st_" + name] = st.astype(np.uint8)
        data["hi_" + name] = hi.astype(np.int32)
        data["lo_" + name] = lo.astype(np.int32)
        data["same_order_" + name] = np.array(np.array_equal(nb, nb_hi))
    data["names"] = np.array(names)
    np.savez_compressed(path, **data)
    print("%d cases, high-level == stable-order low-level in %d; skimage %s numpy %s" %
          (len(names), agree, skimage.__version__, np.__version__))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "watershed_sk.npz"))
