"""Golden vectors from the REFERENCE's own Slice.do_boolean_op, calc_image_density and calc_mask_area
(invesalius/data/slice_.py:1878-1923, 2284-2322), imported from /root/reference and called here.

    python3 tests/golden/make_golden_ref_maskops.py

`transforms.convolve_non_zero` (Rust) under calc_mask_area is bound to oracle/'s restatement; everything else is the
reference's numpy.  The methods run on a plain namespace as `self`; Project() holds the two input masks.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden_ref_dowatershed as M  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main(path):
    tmp_root = os.path.join(ROOT, "gpurun_out", "ref_tmp")
    os.makedirs(tmp_root, exist_ok=True)
    tempfile.tempdir = tmp_root
    os.environ["HOME"] = tmp_root
    M._Finder.ROOTS = tuple(r for r in M._Finder.ROOTS if r != "invesalius_rs")
    native = M._Fake("invesalius_rs._native")
    native.convolve_non_zero = lambda volume, kernel, cval: O.convolve_non_zero(volume, kernel, cval)
    sys.modules["invesalius_rs._native"] = native
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    import invesalius.data.mask as rmask
    from invesalius.data import slice_ as rs
    rng = np.random.default_rng(20261001)
    shape = (7, 9, 11)
    img = rng.integers(-1024, 3072, size=shape).astype(np.int16)
    d = {"img": img}
    masks = []
    for k, (lo, hi) in enumerate(((226, 3071), (-300, 600))):
        m = rmask.Mask()
        m.create_mask(shape)
        m.name, m.threshold_range = "M%d" % k, (lo, hi)
        m.matrix[1:, 1:, 1:] = rng.choice(np.array([0, 1, 2, 253, 254, 255], np.uint8), size=shape)
        m.matrix[1:4, 0, 0] = 1          # slices 0..2 are up to date, the others get thresholded by the calls below
        masks.append(m)
        d["mask%d_in" % k] = np.array(m.matrix)
    captured = []
    self_ = types.SimpleNamespace(matrix=img, spacing=(0.5, 0.75, 2.0), buffer_slices={}, current_mask=masks[0],
                                  _add_mask_into_proj=lambda mk, show=True: captured.append(mk))
    self_.do_threshold_to_a_slice = lambda *a, **k: rs.Slice.do_threshold_to_a_slice(self_, *a, **k)
    self_.do_threshold_to_all_slices = lambda mask=None, target_matrix=None: rs.Slice.do_threshold_to_all_slices(self_, mask, img)
    rs.Project = lambda: types.SimpleNamespace(mask_dict={0: masks[0], 1: masks[1]}, image_versions=[])
    for op in (1, 2, 3, 4):
        rs.Slice.do_boolean_op(self_, op, masks[0], masks[1])
        d["bool_%d" % op] = np.array(captured[-1].matrix)
        d["bool_%d_name" % op] = np.array(captured[-1].name)
    d["mask0_after"], d["mask1_after"] = np.array(masks[0].matrix), np.array(masks[1].matrix)
    d["density"] = np.array(rs.Slice.calc_image_density(self_, masks[0]), np.float64)
    _e = np.memmap(os.path.join(tmp_root, "ones.dat"), dtype=np.uint8, mode="w+", shape=(8, 10, 12)); _e[:] = 1; _e.flush(); del _e
    d["density_empty"] = np.array(rs.Slice.calc_image_density(self_, types.SimpleNamespace(matrix=np.memmap(os.path.join(tmp_root, "ones.dat"), dtype=np.uint8, mode="r+", shape=(8, 10, 12)), threshold_range=(0, 0))), np.float64)
    d["area"] = np.array(rs.Slice.calc_mask_area(self_, masks[1]), np.float64)
    np.savez_compressed(path, **d)
    print(len(d), "arrays:", [str(d["bool_%d_name" % op]) for op in (1, 2, 3, 4)], d["density"], float(d["area"]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_maskops.npz"))
