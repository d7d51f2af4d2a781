"""Golden vectors from the REFERENCE's own WaterShedInteractorStyle.OnBrushRelease (invesalius/data/styles.py:1926-1997): the 2-D
watershed of one slice, three orientations, both algorithms, window/level on and off, overwrite on and off.  Imported from
/root/reference and called on a plain namespace; scikit-image's flood runs through the /opt/conda build.

    python3 tests/golden/make_golden_ref_brush.py
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden_ref_dowatershed as M  # noqa: E402


def main(path):
    tmp_root = os.path.join(ROOT, "gpurun_out", "ref_tmp")
    os.makedirs(tmp_root, exist_ok=True)
    tempfile.tempdir = tmp_root
    os.environ["HOME"] = tmp_root
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    from invesalius.data import styles as rst
    rst.watershed = M.skimage_watershed_proxy
    rst.Publisher.sendMessage = lambda *a, **k: None
    img, am = M.ct_like((20, 40, 44), 19)
    rng = np.random.default_rng(4)
    start = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    start[1:, 1:, 1:] = rng.choice(np.array([0, 0, 0, 1, 2, 253, 254, 255], np.uint8), size=img.shape)
    markers = np.zeros(img.shape, np.uint8)
    z, y, x = am
    markers[z - 1:z + 2, y - 2:y + 3, x - 2:x + 3] = 1
    markers[:, :4, :] = 2          # background strokes that every slice through the object also meets
    markers[:3, :, :] = 2
    markers[:, :, :3] = 2
    d = {"img": img, "markers": markers, "mask_in": start}
    names = []
    k = 0
    for orientation, n in (("AXIAL", int(z)), ("CORONAL", int(y)), ("SAGITAL", int(x))):
        for algorithm in ("Watershed", "Watershed IFT"):
            for use_ww_wl in (True, False):
                for overwrite in ((False, True) if orientation == "AXIAL" else (False,)):
                    mm = np.array(start)
                    sl = types.SimpleNamespace(matrix=img, window_width=400, window_level=300, discard_all_buffers=lambda: None,
                                               current_mask=types.SimpleNamespace(matrix=mm, was_edited=False, modified=lambda *a, **kw: None,
                                                                                  clear_history=lambda: None))
                    self_ = types.SimpleNamespace(orientation=orientation, matrix=markers,
                                                  config=types.SimpleNamespace(algorithm=algorithm, con_2d=4, mg_size=3, use_ww_wl=use_ww_wl),
                                                  viewer=types.SimpleNamespace(slice_=sl, overwrite_mask=overwrite, slice_data=types.SimpleNamespace(number=n)))
                    name = "%s_%d_%s_%d_%d" % (orientation, n, algorithm.replace(" ", ""), use_ww_wl, overwrite)
                    try:
                        rst.WaterShedInteractorStyle.OnBrushRelease(self_, None, None)
                        d["err_" + name] = np.array("")
                    except TypeError as e:
                        d["err_" + name] = np.array("TypeError: %s" % e)
                    d["out_" + name] = mm
                    names.append(name)
                    k += 1
    d["names"] = np.array(names)
    np.savez_compressed(path, **d)
    print(len(names), "brush releases;", [(n, str(d["err_" + n])[:40]) for n in names if str(d["err_" + n])])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_brush_watershed.npz"))
