"""Golden vectors from the REFERENCE's own WaterShedInteractorStyle.expand_watershed (invesalius/data/styles.py:2071-2160): the
whole 3-D watershed tool -- do_threshold_to_all_slices on the current mask, do_watershed (in its worker, run inline here),
and the merge rule of :2147-2152 -- imported from /root/reference and called on a plain namespace.

    python3 tests/golden/make_golden_ref_expand.py

scikit-image's flood runs through the /opt/conda build (see make_golden_ref_dowatershed.py).
"""
import os
import queue
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden_ref_dowatershed as M  # noqa: E402


class _InlineProcess:
    def __init__(self, target, args):
        self.target, self.args = target, args

    def start(self):
        self.target(*self.args)

    def is_alive(self):
        return False


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
    from invesalius.data import slice_ as rslice
    from invesalius.data import styles as rst
    from invesalius.data import watershed_process as rwp
    rwp.watershed = M.skimage_watershed_proxy
    rst.multiprocessing = types.SimpleNamespace(Queue=queue.Queue, Process=_InlineProcess)
    rst.wx.GetMousePosition = lambda: (0, 0)
    rst.Publisher.sendMessage = lambda *a, **k: None
    img, am = M.ct_like((12, 26, 30), 77)
    markers = np.zeros(img.shape, np.uint8)
    markers[max(am[0] - 1, 0):am[0] + 2, am[1] - 2:am[1] + 3, am[2] - 2:am[2] + 3] = 1   # BRUSH_FOREGROUND
    markers[:2, :4, :4] = 2                                                                # BRUSH_BACKGROUND
    markers[-2:, -4:, -4:] = 2
    rng = np.random.default_rng(3)
    start = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    start[1:, 1:, 1:] = rng.choice(np.array([0, 0, 0, 1, 2, 253, 254, 255], np.uint8), size=img.shape)
    start[1:5, 0, 0] = 1     # slices 0..3 already thresholded, the rest is (re)thresholded by the tool first
    d = {"img": img, "markers": markers, "mask_in": start}
    k = 0
    for algorithm in ("Watershed", "Watershed IFT"):
        for use_ww_wl in (True, False):
            for overwrite in (False, True):
                mm = np.memmap(os.path.join(tmp_root, "expand_%d.dat" % k), dtype=np.uint8, mode="w+", shape=start.shape)
                mm[:] = start
                cur = types.SimpleNamespace(matrix=mm, threshold_range=(226, 3071), modified=lambda *a, **kw: None,
                                            clear_history=lambda: None)
                sl = types.SimpleNamespace(matrix=img, current_mask=cur, window_width=400, window_level=300,
                                           discard_all_buffers=lambda: None)
                sl.do_threshold_to_a_slice = lambda *a, **kw: rslice.Slice.do_threshold_to_a_slice(sl, *a, **kw)
                sl.do_threshold_to_all_slices = lambda: rslice.Slice.do_threshold_to_all_slices(sl, cur, img)
                viewer = types.SimpleNamespace(slice_=sl, overwrite_mask=overwrite, ScreenToClient=lambda p: (0, 0),
                                               interactor=types.SimpleNamespace(HitTest=lambda p: 0))
                self_ = types.SimpleNamespace(matrix=markers, viewer=viewer, OnEnterInteractor=lambda *a: None,
                                              config=types.SimpleNamespace(algorithm=algorithm, con_3d=6, mg_size=(3, 3, 3),
                                                                           use_ww_wl=use_ww_wl))
                rst.WaterShedInteractorStyle.expand_watershed(self_)
                name = "%s_%d_%d" % (algorithm.replace(" ", ""), use_ww_wl, overwrite)
                d["out_" + name] = np.array(mm)
                k += 1
    d["names"] = np.array([n[4:] for n in d if n.startswith("out_")])
    np.savez_compressed(path, **d)
    print(len(d["names"]), "runs of the reference's own expand_watershed:", ", ".join(d["names"]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_expand_watershed.npz"))
