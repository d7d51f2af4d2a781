"""Golden vectors from the REFERENCE's own Slice.get_image_slice (invesalius/data/slice_.py:832-1119): slab extraction,
`inverted`, and every projection type with the arguments the reference passes -- window LEVEL for level and width -- through
the real wrappers of invesalius_rs/__init__.py.

    python3 tests/golden/make_golden_ref_mips.py

The Rust under the wrappers (`_native.mida`, `_native.fast_countour_mip`) is bound to oracle/'s C restatement of mips.rs
(unpinned upstream: the reference has no tests for it); MaxIP / MinIP / MeanIP are the reference's own numpy.
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
    native.mida = lambda image, axis, wl, ww, out: O.mida(image, axis, wl, ww, out)
    native.fast_countour_mip = lambda image, n, axis, wl, ww, tmip, out: O.fast_countour_mip(image, n, axis, wl, ww, tmip, out)
    sys.modules["invesalius_rs._native"] = native
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    os.environ.setdefault("HOME", tempfile.mkdtemp())
    sys.path.insert(0, "/root/reference")
    from invesalius.data import slice_ as rs
    rng = np.random.default_rng(20260930)
    from scipy import ndimage
    f = ndimage.gaussian_filter(rng.normal(0, 1, (12, 14, 16)), 1.5)
    img = (f / np.abs(f).max() * 1500 + rng.normal(0, 30, f.shape)).astype(np.int16)
    d = {"img": img}
    cases = []
    for orientation, n0, ns in (("AXIAL", 2, 7), ("CORONAL", 3, 8), ("SAGITAL", 1, 12), ("AXIAL", 9, 6)):
        for tp in (0, 1, 2, 3, 5, 6, 7, 8):
            for inverted in (False, True):
                self_ = types.SimpleNamespace(matrix=img, _type_projection=tp, q_orientation=np.array((1.0, 0, 0, 0)), window_level=300,
                                              window_width=900, spacing=(1.0, 1.0, 1.0), center=(0, 0, 0), interp_method=2,
                                              buffer_slices={orientation: types.SimpleNamespace(index=-1, image=None)})
                out = rs.Slice.get_image_slice(self_, orientation, n0, ns, inverted, 1.0)
                name = "%s_%d_%d_%d_%d" % (orientation, n0, ns, tp, inverted)
                cases.append(name)
                d[name] = np.array(out)
    # Q2: the LMIP branch dies on the missing export
    self_ = types.SimpleNamespace(matrix=img, _type_projection=4, q_orientation=np.array((1.0, 0, 0, 0)), window_level=300,
                                  buffer_slices={"AXIAL": types.SimpleNamespace(index=-1, image=None)})
    try:
        rs.Slice.get_image_slice(self_, "AXIAL", 2, 7, False, 1.0)
        d["lmip_error"] = np.array("none")
    except AttributeError as e:
        d["lmip_error"] = np.array(type(e).__name__)
    d["cases"] = np.array(cases)
    np.savez_compressed(path, **d)
    print(len(cases), "projections by the reference's own get_image_slice; LMIP ->", d["lmip_error"])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_mips.npz"))
