"""Golden vectors from the REFERENCE's own SelectMaskPartsInteractorStyle.OnSelect (invesalius/data/styles.py:2883-2960): two
clicks select two parts of a mask into the selection mask, a Ctrl+click removes one again.  Imported from /root/reference,
called on a plain namespace; the Rust flood under the real wrapper is oracle/'s C restatement.

    python3 tests/golden/make_golden_ref_select.py
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
    native.floodfill_threshold = lambda data, seeds, t0, t1, fill, strct, out: O.floodfill_threshold(data, seeds, t0, t1, fill, strct, out)
    sys.modules["invesalius_rs._native"] = native
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    from scipy import ndimage
    from invesalius.data import slice_ as rslice
    from invesalius.data import styles as rst
    rst.Publisher.sendMessage = lambda *a, **k: None
    img = np.full((10, 24, 26), -1000, np.int16)          # three separate bright parts, one of them touching noise
    rng = np.random.default_rng(63)
    img[2:6, 3:9, 4:12] = 1200
    img[4:9, 13:21, 5:11] = 1500
    img[1:4, 14:20, 16:24] = 1000
    img += rng.integers(-40, 40, size=img.shape).astype(np.int16)
    start = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    start[1:, 1:, 1:] = np.where(img >= 900, 255, 0)   # (several separate parts)
    start[1:, 0, 0] = 1
    lab, n = ndimage.label(start[1:, 1:, 1:] > 127, ndimage.generate_binary_structure(3, 1))
    sizes = np.bincount(lab.ravel())[1:]
    order = np.argsort(-sizes)
    seeds = [tuple(int(v) for v in np.argwhere(lab == order[k] + 1)[0][::-1]) for k in (0, 1)]
    d = {"img": img, "mask_in": start, "seeds": np.array(seeds)}
    mm = np.memmap(os.path.join(tmp_root, "sel_mask.dat"), dtype=np.uint8, mode="w+", shape=start.shape)
    mm[:] = start
    selm = np.memmap(os.path.join(tmp_root, "sel_sel.dat"), dtype=np.uint8, mode="w+", shape=start.shape)
    selm[:] = 0
    cur = types.SimpleNamespace(matrix=mm, threshold_range=(900, 3071), derived_from="original")
    sl = types.SimpleNamespace(matrix=img, current_mask=cur, aux_matrices={}, to_show_aux="", current_image_label="original",
                               buffer_slices={"AXIAL": types.SimpleNamespace(mask=np.zeros((2, 2), np.uint8), index=0, discard_mask=lambda: None,
                                                                            discard_vtk_mask=lambda: None)})
    sl.do_threshold_to_a_slice = lambda *a, **kw: rslice.Slice.do_threshold_to_a_slice(sl, *a, **kw)
    sl.do_threshold_to_all_slices = lambda: rslice.Slice.do_threshold_to_all_slices(sl, cur, img)
    ctrl = {"on": False}
    cfg = types.SimpleNamespace(con_3d=6, seeds=[], mask=types.SimpleNamespace(matrix=selm, was_edited=False, derived_from="original"))
    state = {"seed": seeds[0]}
    self_ = types.SimpleNamespace(orientation="AXIAL", picker=None, t0=253, t1=255, fill_value=254, config=cfg, slice_data=types.SimpleNamespace(number=0),
                                  GetMousePosition=lambda: (0, 0),
                                  viewer=types.SimpleNamespace(slice_=sl, interactor=types.SimpleNamespace(GetControlKey=lambda: ctrl["on"], Render=lambda: None),
                                                               get_voxel_coord_by_screen_pos=lambda mx, my, pk: state["seed"]))
    steps = []
    for k, (seed, c) in enumerate(((seeds[0], False), (seeds[1], False), (seeds[0], True))):
        state["seed"], ctrl["on"] = seed, c
        try:
            rst.SelectMaskPartsInteractorStyle.OnSelect(self_, None, None)
        except Exception as e:  # whatever the GUI tail of the handler wants after the array work
            print("after the array work:", type(e).__name__, e)
        d["sel_%d" % k] = np.array(selm)
        steps.append(int((selm[1:, 1:, 1:] == 254).sum()))
    d["mask_after"] = np.array(mm)
    np.savez_compressed(path, **d)
    print("selected voxels after each click:", steps)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_select.npz"))
