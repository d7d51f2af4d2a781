"""Golden vectors from the REFERENCE's own Slice.apply_reorientation (invesalius/data/slice_.py:1969-2068): the method is
called unbound on a plain namespace that carries what it reads (matrix memmap, spacing, center, q_orientation, interp_method,
buffer_slices) with `Project()` answering a stand-in project whose masks are real np.memmaps -- one EDITED mask (resampled
with nearest neighbour, padded matrix and all) and one threshold mask (cleared).

    python3 tests/golden/make_golden_ref_applyreorient.py

`_native.apply_view_matrix_transform` is bound to oracle/'s C restatement of transforms.rs / interpolation.rs (unpinned
upstream: no Rust toolchain here, no tests there); everything above it -- the matrix from transformations.py, the copy, the
argument order, cval = copy.min(), which masks are resampled / cleared and how, the view state left behind -- is the
reference's own code.
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


def _memmap(a, d, name):
    m = np.memmap(os.path.join(d, name), shape=a.shape, dtype=a.dtype, mode="w+")
    m[:] = a
    return m


def main(path):
    M._Finder.ROOTS = tuple(r for r in M._Finder.ROOTS if r != "invesalius_rs")
    native = M._Fake("invesalius_rs._native")
    calls = []

    def avmt(volume, spacing, m, n, orientation, minterpol, cval, out):
        calls.append((str(volume.dtype), int(n), orientation, int(minterpol), float(cval)))
        O.apply_view_matrix_transform(np.ascontiguousarray(volume), spacing, m, n, orientation, minterpol, cval, out)

    native.apply_view_matrix_transform = avmt
    sys.modules["invesalius_rs._native"] = native
    sys.meta_path.insert(0, M._Finder())
    import pubsub.pub
    pubsub.pub.subscribe = lambda *a, **k: (None, True)
    pubsub.pub.sendMessage = lambda *a, **k: None
    os.environ.setdefault("HOME", tempfile.mkdtemp())
    sys.path.insert(0, "/root/reference")
    from invesalius.data import slice_ as rs
    rng = np.random.default_rng(20261001)
    from scipy import ndimage
    f = ndimage.gaussian_filter(rng.normal(0, 1, (11, 13, 17)), 1.5)
    img = (f / np.abs(f).max() * 1500 + rng.normal(0, 30, f.shape)).astype(np.int16)
    spacing = (0.5, 0.75, 1.25)
    center = [(s * d / 2.0) for (d, s) in zip(img.shape[::-1], spacing)]
    quats = [np.array((0.9238795325112867, 0.3826834323650898, 0.0, 0.0)), np.array((0.8, 0.2, -0.4, 0.4)),
             np.array((0.0, 0.0, 1.0, 0.0))]
    edited = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    edited[1:, 1:, 1:] = np.where(img > 200, 255, 0)
    edited[4:8, 5:9, 6:11] = rng.choice(np.array([0, 1, 2, 253, 254, 255], np.uint8), (4, 4, 5))  # brush / watershed values
    edited[1:, 0, 0] = 2          # flag cells of an edited mask (slice_.py:1940-1960)
    edited[0, 1:, 0] = 2
    edited[0, 0, 1:] = 2
    thresholded = np.zeros_like(edited)
    thresholded[1:, 1:, 1:] = np.where(img > 0, 255, 0)
    thresholded[1:, 0, 0] = 1
    d = {"img": img, "spacing": np.array(spacing), "center": np.array(center), "edited": edited, "thresholded": thresholded}
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for qi, q in enumerate(quats):
            d["q%d" % qi] = q
            for interp in (0, 1, 2, 3):
                calls.clear()
                masks = {}
                for mi, (arr, was_edited) in enumerate(((edited, True), (thresholded, False))):
                    fd, name = tempfile.mkstemp(dir=tmp)
                    mm = np.memmap(name, shape=arr.shape, dtype=np.uint8, mode="w+")
                    mm[:] = arr
                    log = []
                    masks[mi] = types.SimpleNamespace(matrix=mm, was_edited=was_edited, temp_fd=fd, clear_history=lambda log=log: log.append(1),
                                                      _recreate_mask_matrix=lambda shape: (_ for _ in ()).throw(AssertionError("shape changed")),
                                                      log=log)
                proj = types.SimpleNamespace(image_versions=[], mask_dict=masks)
                rs.Project = lambda proj=proj: proj
                disc = []
                self_ = types.SimpleNamespace(matrix=_memmap(img, tmp, "m_%d_%d.dat" % (qi, interp)), spacing=spacing, center=list(center),
                                              q_orientation=q, interp_method=interp,
                                              buffer_slices={o: types.SimpleNamespace(discard_buffer=lambda o=o: disc.append(o))
                                                             for o in ("AXIAL", "CORONAL", "SAGITAL")})
                rs.Slice.apply_reorientation(self_)
                name = "%d_%d" % (qi, interp)
                cases.append(name)
                d["img_" + name] = np.array(self_.matrix)
                d["edited_" + name] = np.array(masks[0].matrix)
                d["thresholded_" + name] = np.array(masks[1].matrix)
                d["q_after_" + name] = np.array(self_.q_orientation)
                d["center_after_" + name] = np.array(self_.center)
                assert [c[3] for c in calls] == [interp, 0] and len(masks[0].log) == len(masks[1].log) == 1 and len(disc) == 3, calls
                for m in masks.values():
                    os.close(m.temp_fd)
    d["cases"] = np.array(cases)
    np.savez_compressed(path, **d)
    print(len(cases), "volumes + masks reoriented by the reference's own Slice.apply_reorientation")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_applyreorient.npz"))
