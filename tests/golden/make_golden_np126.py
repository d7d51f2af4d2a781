"""The numpy / scipy expressions of the reference, evaluated under the library versions it pins.

    /opt/conda/bin/python3.9 tests/golden/make_golden_np126.py      # numpy 1.26.4 (the reference's pin), scipy 1.7.1

oracle/oracle.py restates a1-a3, a10-a12 of SURVEY 8 (threshold, LUT, merge rule, projections) as the reference's own numpy
expressions; the test interpreter runs numpy 2.2 (other scalar-promotion rules, NEP 50).  This script imports that same
oracle module under numpy 1.26.4 and stores what it returns, so the tests can show that the restatement does not depend
on the numpy generation -- and the scipy calls of the reference (watershed_ift, morphological_gradient, zoom(order=2),
label) under the older scipy that sits next to it.  Nothing but the .npz is read at test time.
"""
import os
import sys

import numpy as np
import scipy
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def main(path):
    rng = np.random.default_rng(20260926)
    d = {"versions": np.array([np.__version__, scipy.__version__])}
    img = rng.integers(-1024, 3072, size=(6, 9, 11)).astype(np.int16)
    d["img"] = img
    # a1 / a2: preserve rule (1, 2, 253, 254 survive), flag column
    mask = np.zeros((7, 10, 12), np.uint8)
    mask[1:, 1:, 1:] = rng.choice(np.array([0, 1, 2, 253, 254, 255], np.uint8), size=img.shape)
    mask[3, 0, 0] = 1
    d["mask_in"] = mask.copy()
    m = mask.copy()
    O.do_threshold_to_all_slices(m, img, (226, 3071))
    d["mask_all_slices"] = m
    d["a_slice"] = O.do_threshold_to_a_slice(img[2], mask[3, 1:, 1:], (226, 3071))
    m = mask.copy()
    O.set_mask_threshold_volume(m, img, (-200, 500))
    d["mask_set_threshold"] = m
    d["slice_preview"] = O.set_mask_threshold_slice(img[1], (-200, 500))
    # a11: np.piecewise keeps the input dtype (int16 in -> int16 out, truncation toward zero)
    for i, (w, l) in enumerate([(400, 300), (2000, 500), (1, 0), (255, 127)]):
        d["lut_%d" % i] = O.get_LUT_value(img, w, l)
        d["lut255_%d" % i] = O.get_LUT_value_255(img, w, l)
    # a10: merge rule
    lab = rng.integers(0, 3, size=img.shape).astype(np.uint8)
    d["lab"] = lab
    for ow in (0, 1):
        mm = mask[1:, 1:, 1:].copy()
        O.watershed_merge(mm, lab, bool(ow))
        d["merge_%d" % ow] = mm
    # a12
    for ax in range(3):
        d["max_%d" % ax], d["min_%d" % ax], d["mean_%d" % ax] = O.maxip(img, ax), O.minip(img, ax), O.meanip(img, ax)
    # the scipy calls of the reference under the older scipy
    cost = (img - img.min()).astype("uint16")
    d["minshift"] = cost
    d["grad3"] = ndimage.morphological_gradient(cost, (3, 3, 3))
    mk = np.zeros(img.shape, np.int8)
    mk[2, 4, 5] = 1
    mk[0, 0, 0] = 2
    d["mk"] = mk
    for c in (1, 2, 3):
        d["ift_%d" % c] = ndimage.watershed_ift(cost, mk, ndimage.generate_binary_structure(3, c))
    vol = rng.integers(-1000, 2000, size=(9, 10, 11)).astype(np.int16)
    d["zoom_in"] = vol
    for i, f in enumerate((0.5, 0.75)):
        d["zoom_%d" % i] = ndimage.zoom(vol, f, vol.dtype, order=2)
    bw = (rng.random((8, 9, 10)) < 0.45).astype(np.uint8)
    d["bw"] = bw
    for c in (1, 2, 3):
        d["label_%d" % c], _ = ndimage.label(bw, ndimage.generate_binary_structure(3, c))
    # a8: confidence-connected growing (float64 mean / std of the grown set decide the next thresholds)
    smooth = ndimage.gaussian_filter(rng.normal(0, 1, (12, 20, 22)), 2.0)
    cimg = (smooth / np.abs(smooth).max() * 900 + rng.normal(0, 20, smooth.shape)).astype(np.int16)
    d["conf_img"] = cimg
    seed = tuple(int(v) for v in np.unravel_index(int(np.argmax(smooth)), smooth.shape)[::-1])
    d["conf_seed"] = np.array(seed)
    for c in (1, 3):
        d["conf_%d" % c] = O.do_rg_confidence(cimg, seed, ndimage.generate_binary_structure(3, c), 2.5, 3)
    # Mask.fill_holes_auto (scipy label + the restated Rust fill)
    fm = np.zeros((9, 14, 15), np.uint8)
    fm[1:, 1:, 1:] = np.where(rng.random((8, 13, 14)) < 0.7, 255, 0)
    d["holes_in"] = fm
    for conn in (6, 18, 26):
        t = fm.copy()
        O.mask_fill_holes_auto(t, "3D", conn, "AXIAL", 0, 4)
        d["holes_%d" % conn] = t
    np.savez_compressed(path, **d)
    print("numpy", np.__version__, "scipy", scipy.__version__, len(d), "arrays")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "np126.npz"))
