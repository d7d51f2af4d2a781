"""Golden vectors from the REFERENCE's own FloodFillSegmentInteractorStyle.do_3d_seg (invesalius/data/styles.py:3151-3218), the
3-D region-growing tool: click -> threshold / dynamic / confidence flood -> mask[out] = fill value, after the threshold of the
stale slices.  Imported from /root/reference and called on a plain namespace; the Rust flood under the real invesalius_rs
wrappers is bound to oracle/'s pinned C restatement.

    python3 tests/golden/make_golden_ref_3dseg.py
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
    native.floodfill_threshold = lambda data, seeds, t0, t1, fill, strct, out: O.floodfill_threshold(data, seeds, t0, t1, fill, strct, out)
    sys.modules["invesalius_rs._native"] = native
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    from invesalius.data import slice_ as rslice
    from invesalius.data import styles as rst
    img, am = M.ct_like((12, 28, 30), 91)
    seed = (int(am[2]), int(am[1]), int(am[0]))
    rng = np.random.default_rng(8)
    start = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    start[1:, 1:, 1:] = rng.choice(np.array([0, 0, 1, 2, 253, 254, 255], np.uint8), size=img.shape)
    start[1:4, 0, 0] = 1
    d = {"img": img, "seed": np.array(seed), "mask_in": start}
    cases = [("threshold", False, 6, dict(t0=500, t1=3000)), ("threshold", False, 26, dict(t0=-2000, t1=-1500)),   # second: click rejected
             ("dynamic", False, 6, dict(dev_min=400, dev_max=300)), ("dynamic", True, 18, dict(dev_min=60, dev_max=40)),
             ("confidence", False, 6, {}), ("confidence", True, 26, {})]
    names = []
    for k, (method, use_ww_wl, con, extra) in enumerate(cases):
        mm = np.memmap(os.path.join(tmp_root, "seg_%d.dat" % k), dtype=np.uint8, mode="w+", shape=start.shape)
        mm[:] = start
        cur = types.SimpleNamespace(matrix=mm, threshold_range=(226, 3071), save_history=lambda *a, **kw: None)
        sl = types.SimpleNamespace(matrix=img, current_mask=cur, window_width=900, window_level=400)
        sl.do_threshold_to_a_slice = lambda *a, **kw: rslice.Slice.do_threshold_to_a_slice(sl, *a, **kw)
        sl.do_threshold_to_all_slices = lambda: rslice.Slice.do_threshold_to_all_slices(sl, cur, img)
        cfg = types.SimpleNamespace(method=method, use_ww_wl=use_ww_wl, con_3d=con, fill_value=254, confid_mult=2.5, confid_iters=3,
                                    t0=0, t1=0, dev_min=0, dev_max=0, dlg=mock.MagicMock())
        for kk, vv in extra.items():
            setattr(cfg, kk, vv)
        self_ = types.SimpleNamespace(config=cfg, picker=None, GetMousePosition=lambda: (0, 0),
                                      viewer=types.SimpleNamespace(slice_=sl, get_voxel_coord_by_screen_pos=lambda mx, my, pk: seed))
        self_.do_rg_confidence = lambda *a, **kw: rst.FloodFillSegmentInteractorStyle.do_rg_confidence(self_, *a, **kw)
        rst.FloodFillSegmentInteractorStyle.do_3d_seg(self_)
        name = "%s_%d_%d_%d" % (method, use_ww_wl, con, k)
        names.append(name)
        d["out_" + name] = np.array(mm)
        d["cfg_" + name] = np.array([cfg.t0, cfg.t1, cfg.dev_min, cfg.dev_max])
    d["names"] = np.array(names)
    np.savez_compressed(path, **d)
    print(len(names), "clicks through the reference's own do_3d_seg:",
          [(n, int((d["out_" + n] != start).sum())) for n in names])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_3dseg.npz"))
