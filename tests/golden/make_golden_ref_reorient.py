"""Golden vectors from the REFERENCE's own Slice.get_image_slice with a REORIENTED view (invesalius/data/slice_.py:844-874,
948-958, 1035-1045): the 4x4 matrix it builds with invesalius/data/transformations.py (T1 . R^T . T0 from q_orientation and
center), the slab it resamples with transforms.apply_view_matrix_transform, then `inverted` and the projection.

    python3 tests/golden/make_golden_ref_reorient.py

`_native.apply_view_matrix_transform`, `_native.mida` and `_native.fast_countour_mip` are bound to oracle/'s C restatements
of transforms.rs / interpolation.rs / mips.rs (unpinned upstream: no Rust toolchain here, no tests there); everything above
them -- the matrix, the argument order, the slab bookkeeping, the numpy projections -- is the reference's own code.
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
    mats = []

    def avmt(volume, spacing, m, n, orientation, minterpol, cval, out):
        mats.append(np.array(m, dtype=np.float64))
        O.apply_view_matrix_transform(volume, spacing, m, n, orientation, minterpol, cval, out)

    native.apply_view_matrix_transform = avmt
    sys.modules["invesalius_rs._native"] = native
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    os.environ.setdefault("HOME", tempfile.mkdtemp())
    sys.path.insert(0, "/root/reference")
    from invesalius.data import slice_ as rs
    rng = np.random.default_rng(20260931)
    from scipy import ndimage
    f = ndimage.gaussian_filter(rng.normal(0, 1, (12, 14, 16)), 1.5)
    img = (f / np.abs(f).max() * 1500 + rng.normal(0, 30, f.shape)).astype(np.int16)
    spacing = (0.5, 0.75, 1.25)
    center = [(s * d / 2.0) for (d, s) in zip(img.shape[::-1], spacing)]  # slice_.py:2003
    quats = [np.array((0.9238795325112867, 0.3826834323650898, 0.0, 0.0)),   # 45 degrees about one axis
             np.array((0.8, 0.2, -0.4, 0.4)),                                  # a general rotation (not normalised)
             np.array((0.0, 0.0, 1.0, 0.0))]                                   # a half turn
    d = {"img": img, "spacing": np.array(spacing), "center": np.array(center)}
    cases = []
    for qi, q in enumerate(quats):
        d["q%d" % qi] = q
        for orientation, n0, ns in (("AXIAL", 2, 5), ("CORONAL", 3, 6), ("SAGITAL", 4, 7)):
            for interp in (0, 1, 2, 3):
                for tp, inverted in ((0, False), (1, False), (3, True), (5, False), (6, True)):
                    if interp in (2, 3) and tp in (3, 6):
                        continue  # (keep the file small: the slow kernels get the two cheapest projections)
                    mats.clear()
                    self_ = types.SimpleNamespace(matrix=img, _type_projection=tp, q_orientation=q, window_level=300, window_width=900,
                                                  spacing=spacing, center=center, interp_method=interp,
                                                  buffer_slices={orientation: types.SimpleNamespace(index=-1, image=None)})
                    out = rs.Slice.get_image_slice(self_, orientation, n0, ns, inverted, 1.0)
                    name = "%d_%s_%d_%d_%d_%d_%d" % (qi, orientation, n0, ns, interp, tp, inverted)
                    cases.append(name)
                    d[name] = np.array(out)
                    assert len(mats) == 1
                    d["M%d" % qi] = mats[0]
    d["cases"] = np.array(cases)
    np.savez_compressed(path, **d)
    print(len(cases), "reoriented slabs by the reference's own get_image_slice")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_reorient.npz"))
