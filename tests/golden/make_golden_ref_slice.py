"""Golden vectors from the REFERENCE ITSELF for the numpy stages: Slice.do_threshold_to_a_slice / do_threshold_to_all_slices /
SetMaskThreshold (invesalius/data/slice_.py:1225-1267,1722-1769), get_LUT_value / get_LUT_value_255 and resize_image_array
(invesalius/data/imagedata_utils.py:121-130,540-564), imported from /root/reference and called here.

    python3 tests/golden/make_golden_ref_slice.py                # Python 3.10, stand-in modules for the GUI-side imports

The Slice methods are called unbound with a plain namespace as `self` (they use nothing of the singleton beyond the
attributes set below); Project() is replaced by a namespace that holds the one mask.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ref_dowatershed as M  # noqa: E402  (the stand-in module finder)


def main(path):
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    os.environ.setdefault("HOME", tempfile.mkdtemp())
    sys.path.insert(0, "/root/reference")
    from invesalius.data import imagedata_utils as iu
    from invesalius.data import slice_ as rs
    rs.Publisher.sendMessage = lambda *a, **k: None
    rng = np.random.default_rng(20260927)
    d = {}
    img = rng.integers(-1024, 3072, size=(7, 10, 12)).astype(np.int16)
    d["img"] = img
    tmpd = tempfile.mkdtemp()

    def new_mask(fill):
        mm = np.memmap(os.path.join(tmpd, "m%d.dat" % rng.integers(1 << 30)), shape=(8, 11, 13), dtype=np.uint8, mode="w+")
        mm[:] = fill
        return mm

    start = np.zeros((8, 11, 13), np.uint8)
    start[1:, 1:, 1:] = rng.choice(np.array([0, 1, 2, 253, 254, 255], np.uint8), size=img.shape)
    start[2, 0, 0] = 1   # slice 1 already thresholded: do_threshold_to_all_slices leaves it alone
    start[5, 0, 0] = 2   # edited
    d["mask_in"] = start
    self_ = types.SimpleNamespace()
    self_.do_threshold_to_a_slice = lambda *a, **k: rs.Slice.do_threshold_to_a_slice(self_, *a, **k)
    # a1
    d["a_slice"] = rs.Slice.do_threshold_to_a_slice(self_, img[3], start[4, 1:, 1:], (226, 3071))
    self_.current_mask = types.SimpleNamespace(threshold_range=(-100, 400))
    d["a_slice_current"] = rs.Slice.do_threshold_to_a_slice(self_, img[3], start[4, 1:, 1:])
    # a2
    mask = types.SimpleNamespace(matrix=new_mask(start), threshold_range=(226, 3071))
    rs.Slice.do_threshold_to_all_slices(self_, mask=mask, target_matrix=img)
    d["all_slices"] = np.array(mask.matrix)
    # a3: whole volume, and the per-slice preview
    cur = types.SimpleNamespace(matrix=new_mask(start), threshold_range=(0, 0), was_edited=True)
    self_.current_mask = cur
    self_.matrix = img
    rs.Project = lambda: types.SimpleNamespace(mask_dict={0: cur})
    rs.Slice.SetMaskThreshold(self_, 0, (-200, 500))
    d["set_threshold_volume"] = np.array(cur.matrix)
    self_.buffer_slices = {"AXIAL": types.SimpleNamespace(image=img[2], mask=None)}
    rs.Slice.SetMaskThreshold(self_, 0, (-200, 500), slice_number=2, orientation="AXIAL")
    d["set_threshold_preview"] = self_.buffer_slices["AXIAL"].mask
    # a11
    wl = [(400, 300), (2000, 500), (1, 0), (255, 127), (80, 40)]
    d["wl"] = np.array(wl)
    for i, (w, l) in enumerate(wl):
        d["lut_%d" % i] = iu.get_LUT_value(img, w, l)
        d["lut255_%d" % i] = iu.get_LUT_value_255(img, w, l)
    # a17
    vol = rng.integers(-1000, 2000, size=(11, 12, 13)).astype(np.int16)
    m8 = rng.choice(np.array([0, 255], np.uint8), size=(11, 12, 13))
    d["zoom_i16"], d["zoom_u8"] = vol, m8
    for i, f in enumerate((0.5, 0.75)):
        d["zoom_i16_%d" % i] = iu.resize_image_array(vol, f)
        d["zoom_u8_%d" % i] = iu.resize_image_array(m8, f)
    np.savez_compressed(path, **d)
    print(len(d), "arrays from the reference's own functions")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_slice.npz"))
