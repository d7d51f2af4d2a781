"""Golden vectors from the REFERENCE's own FloodFillMaskInteractorStyle.OnFFClick / RemoveMaskPartsInteractorStyle (invesalius/
data/styles.py:2434-2589): the "fill holes" and "remove parts" tools, 3-D and inside one slice of each orientation.  Imported
from /root/reference, called on a plain namespace; the Rust flood under the real wrapper is oracle/'s C restatement.

    python3 tests/golden/make_golden_ref_ffmask.py
"""
import os
import sys
import tempfile
import types
from unittest import mock

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
    native.floodfill_threshold_inplace = lambda data, seeds, t0, t1, fill, strct: O.floodfill_threshold_inplace(data, seeds, t0, t1, fill, strct)
    sys.modules["invesalius_rs._native"] = native
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    from invesalius.data import slice_ as rslice
    from invesalius.data import styles as rst
    rst.Publisher.sendMessage = lambda *a, **k: None
    rst.wx.ProgressDialog = lambda *a, **k: mock.MagicMock()
    img, am = M.ct_like((10, 24, 26), 55)
    rng = np.random.default_rng(12)
    start = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    start[1:, 1:, 1:] = np.where(img >= 226, 255, 0)
    from scipy import ndimage
    field = ndimage.gaussian_filter(rng.normal(0, 1, img.shape), 1.2)
    holes = field > np.quantile(field, 0.75)   # connected pockets, not single voxels
    start[1:, 1:, 1:][holes & (img >= 226)] = rng.choice(np.array([0, 1, 2], np.uint8), size=int((holes & (img >= 226)).sum()))
    start[1:, 1:, 1:][rng.random(img.shape) < 0.03] = 254
    start[1:4, 0, 0] = 1
    d = {"img": img, "mask_in": start}
    inner = start[1:, 1:, 1:]
    hole = tuple(int(v) for v in np.argwhere(inner <= 2)[len(np.argwhere(inner <= 2)) // 2][::-1])
    part = tuple(int(v) for v in np.argwhere(inner >= 253)[len(np.argwhere(inner >= 253)) // 2][::-1])
    cases = [("fill", "3D", "AXIAL", 6, 4, hole), ("fill", "3D", "AXIAL", 26, 4, hole), ("fill", "2D", "AXIAL", 6, 4, hole),
             ("fill", "2D", "CORONAL", 6, 8, hole), ("fill", "2D", "SAGITAL", 6, 4, hole), ("fill", "3D", "AXIAL", 6, 4, part),  # rejected
             ("remove", "3D", "AXIAL", 18, 4, part), ("remove", "2D", "CORONAL", 6, 4, part), ("remove", "2D", "AXIAL", 6, 8, hole)]  # last: rejected
    names = []
    for k, (tool, target, orientation, c3, c2, seed) in enumerate(cases):
        mm = np.memmap(os.path.join(tmp_root, "ff_%d.dat" % k), dtype=np.uint8, mode="w+", shape=start.shape)
        mm[:] = start
        cur = types.SimpleNamespace(matrix=mm, threshold_range=(226, 3071), save_history=lambda *a, **kw: None, modified=lambda *a, **kw: None,
                                    was_edited=False)
        buf = {o: types.SimpleNamespace(mask=np.zeros((2, 2), np.uint8), index=seed[{"AXIAL": 2, "CORONAL": 1, "SAGITAL": 0}[o]],
                                        discard_mask=lambda: None, discard_vtk_mask=lambda: None) for o in ("AXIAL", "CORONAL", "SAGITAL")}
        sl = types.SimpleNamespace(matrix=img, current_mask=cur, buffer_slices=buf)
        sl.do_threshold_to_a_slice = lambda *a, **kw: rslice.Slice.do_threshold_to_a_slice(sl, *a, **kw)
        sl.do_threshold_to_all_slices = lambda: rslice.Slice.do_threshold_to_all_slices(sl, cur, img)
        t0, t1, fill = (0, 2, 254) if tool == "fill" else (253, 255, 1)
        self_ = types.SimpleNamespace(orientation=orientation, picker=None, t0=t0, t1=t1, fill_value=fill, slice_data=types.SimpleNamespace(number=0),
                                      GetMousePosition=lambda: (0, 0), _progr_title="", _progr_msg="",
                                      config=types.SimpleNamespace(target=target, con_3d=c3, con_2d=c2),
                                      viewer=types.SimpleNamespace(slice_=sl, get_voxel_coord_by_screen_pos=lambda mx, my, pk, s=seed: s))
        rst.FloodFillMaskInteractorStyle.OnFFClick(self_, None, None)
        name = "%s_%s_%s_%d_%d_%d" % (tool, target, orientation, c3, c2, k)
        names.append(name)
        d["out_" + name] = np.array(mm)
        d["seed_" + name] = np.array(seed)
    d["names"] = np.array(names)
    np.savez_compressed(path, **d)
    print(len(names), "clicks:", [(n, int((d["out_" + n] != start).sum())) for n in names])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_ffmask.npz"))
