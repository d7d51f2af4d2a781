"""Golden vectors from the REFERENCE's own Python around its native flood fill: FloodFillSegmentInteractorStyle.do_rg_confidence
(invesalius/data/styles.py:3220-3251), Mask.fill_holes_auto (invesalius/data/mask.py:519-562) and the wrappers of
invesalius_rs/__init__.py:21-54, imported from /root/reference and called here.

    python3 tests/golden/make_golden_ref_rg.py

The compiled Rust module (`invesalius_rs._native`) cannot be built in this container (no cargo); its three flood entry points
are bound to oracle/'s C restatement of floodfill.rs -- the one the reference's own golden vectors pin
(tests/test_oracle_golden.py) -- so what these vectors add is the reference's PYTHON on top of it: the truncation rules of
the wrappers, the statistics / threshold loop of the confidence-connected growing (quirk Q4 included), the label / reshape /
size logic of the hole filling.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden_ref_dowatershed as M  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main(path):
    M._Finder.ROOTS = tuple(r for r in M._Finder.ROOTS if r != "invesalius_rs")
    native = M._Fake("invesalius_rs._native")
    native.floodfill_threshold = lambda data, seeds, t0, t1, fill, strct, out: O.floodfill_threshold(data, seeds, t0, t1, fill, strct, out)
    native.floodfill_threshold_inplace = lambda data, seeds, t0, t1, fill, strct: O.floodfill_threshold_inplace(data, seeds, t0, t1, fill, strct)
    native.fill_holes_automatically = lambda mask, labels, nlabels, size: O.fill_holes_automatically(mask, labels, nlabels, size)
    sys.modules["invesalius_rs._native"] = native
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    os.environ.setdefault("HOME", tempfile.mkdtemp())
    sys.path.insert(0, "/root/reference")
    import invesalius_rs as ref_rs
    from scipy import ndimage
    from invesalius.data import mask as ref_mask
    from invesalius.data import styles as ref_styles
    assert ref_rs.__file__.startswith("/root/reference/") and ref_styles.floodfill is ref_rs
    rng = np.random.default_rng(20260928)
    d = {}
    smooth = ndimage.gaussian_filter(rng.normal(0, 1, (14, 22, 24)), 2.0)
    img = (smooth / np.abs(smooth).max() * 900 + rng.normal(0, 20, smooth.shape)).astype(np.int16)
    seed = tuple(int(v) for v in np.unravel_index(int(np.argmax(smooth)), smooth.shape)[::-1])
    d["img"], d["seed"] = img, np.array(seed)
    # a8, raw image and through the LUT (styles.py:3222-3225)
    for conn in (1, 3):
        for use_ww_wl in (False, True):
            self_ = types.SimpleNamespace(config=types.SimpleNamespace(use_ww_wl=use_ww_wl, confid_mult=2.5, confid_iters=3),
                                          viewer=types.SimpleNamespace(slice_=types.SimpleNamespace(window_width=900, window_level=200)))
            out = ref_styles.FloodFillSegmentInteractorStyle.do_rg_confidence(self_, img, np.zeros(img.shape, np.uint8), seed,
                                                                               ndimage.generate_binary_structure(3, conn))
            d["conf_%d_%d" % (conn, use_ww_wl)] = out
    # the wrappers: float thresholds on an int16 image are truncated (21-40); the in-place form passes them through (43-54)
    o = np.zeros(img.shape, np.uint8)
    ref_rs.floodfill_threshold(img, [list(seed)], 299.9, 1500.7, 1.9, ndimage.generate_binary_structure(3, 2), o)
    d["wrap_out"] = o
    m = np.zeros((15, 23, 25), np.uint8)
    m[1:, 1:, 1:] = np.where(img > 300, 255, rng.choice(np.array([0, 1, 2], np.uint8), size=img.shape))
    d["inplace_in"] = m.copy()
    view = m[1:, 1:, 1:]
    z0 = np.argwhere(view == 255)[0]
    ref_rs.floodfill_threshold_inplace(view, [(int(z0[2]), int(z0[1]), int(z0[0]))], 253, 255, 1, ndimage.generate_binary_structure(3, 1))
    d["inplace_seed"] = np.array([int(z0[2]), int(z0[1]), int(z0[0])])
    d["inplace_out"] = m
    # Mask.fill_holes_auto, 3-D and per slice
    fm = np.zeros((9, 14, 15), np.uint8)
    fm[1:, 1:, 1:] = np.where(rng.random((8, 13, 14)) < 0.7, 255, 0)
    d["holes_in"] = fm
    for target, conn, orientation, index, size in (("3D", 6, "AXIAL", 0, 4), ("3D", 26, "AXIAL", 0, 50), ("2D", 4, "AXIAL", 3, 3),
                                                   ("2D", 8, "CORONAL", 5, 6), ("2D", 4, "SAGITAL", 7, 2)):
        self_ = types.SimpleNamespace(matrix=fm.copy(), save_history=lambda *a, **k: None)
        ref_mask.Mask.fill_holes_auto(self_, target, conn, orientation, index, size)
        d["holes_%s_%d_%s_%d_%d" % (target, conn, orientation, index, size)] = self_.matrix
    np.savez_compressed(path, **d)
    print(len(d), "arrays from the reference's own Python over the restated native flood")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_rg.npz"))
