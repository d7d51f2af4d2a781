"""``invesalius.data.watershed_process`` on the GPU -- the deterministic stages.

``do_watershed`` (watershed_process.py:19-60) is: [window/level LUT | subtract the minimum] -> uint16 ->
[morphological gradient] -> marker flood (``skimage.segmentation.watershed`` or ``scipy.ndimage.watershed_ift``)
-> uint8 labels, followed in the caller by the merge rule (styles.py:2147-2152).

What runs in HIP kernels here: the LUT / min-shift, the 3x3x3 (or larger, odd) morphological gradient and the
merge.  The marker flood itself is a strictly sequential priority flood whose tie-breaking (LIFO buckets in
scipy, (value, age) heap in scikit-image) decides the labels on every plateau; a bit-exact parallel
formulation is not built yet, so ``do_watershed`` hands the GPU-made cost image to the
very same third-party function the reference calls.  That step is NOT part of libivx and is excluded from every
parity / performance claim.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib as L


def cost_image(image: np.ndarray, use_ww_wl: bool, wl, ww, gradient_size=0) -> np.ndarray:
    """uint16 flood input: ``get_LUT_value(image, ww, wl).astype("uint16")`` or
    ``(image - image.min()).astype("uint16")`` (watershed_process.py:34,42,47,55), then optionally
    ``ndimage.morphological_gradient(tmp, gradient_size)`` (watershed_process.py:36-38,49-51)."""
    if image.dtype != np.int16 or image.ndim not in (2, 3):
        raise TypeError("image must be a 2-D or 3-D int16 array")
    img3 = image if image.ndim == 3 else image[np.newaxis]
    out = np.empty(img3.shape, np.uint16)
    gs = None
    if gradient_size:  # int -> same size on every axis of the input (scipy semantics); tuple -> per axis
        sz = tuple(int(v) for v in gradient_size) if np.ndim(gradient_size) else (int(gradient_size),) * image.ndim
        if len(sz) != image.ndim:
            raise RuntimeError("size must have one entry per image axis")
        sz = (1,) * (3 - len(sz)) + sz
        gs = (ctypes.c_int * 3)(*sz)
    L.check(L.lib().ivx_watershed_prepare(L.ptr(img3), L.i64(img3.shape), L.i64(img3.strides), int(bool(use_ww_wl)),
                                          ctypes.c_double(float(ww)), ctypes.c_double(float(wl)), gs,
                                          L.ptr(out)), "watershed cost image")
    return out.reshape(image.shape)


def merge(mask: np.ndarray, tmp_mask: np.ndarray, overwrite: bool):
    """styles.py:2147-2152 (3-D) / 1984-1989 (2-D), in place on `mask` (a strided view is fine)."""
    if mask.dtype != np.uint8 or tmp_mask.dtype != np.uint8 or mask.shape != tmp_mask.shape:
        raise TypeError("mask and tmp_mask must be uint8 arrays of the same shape")
    m3 = mask if mask.ndim == 3 else mask[np.newaxis]
    t3 = tmp_mask if tmp_mask.ndim == 3 else tmp_mask[np.newaxis]
    L.check(L.lib().ivx_watershed_merge(L.ptr(m3), L.i64(m3.shape), L.i64(m3.strides), L.ptr(t3), L.i64(t3.strides),
                                        int(bool(overwrite))), "watershed merge")


def do_watershed(image, markers, tfile, shape, bstruct, algorithm, mg_size, use_ww_wl, wl, ww, q=None, flood=None):
    """Same signature and side effects as watershed_process.do_watershed (:19-60): writes the uint8 label volume to
    the memmap `tfile` and signals ``q.put(1)``.  The cost image is made on the GPU.

    The marker flood is NOT implemented in libivx (module docstring): by default this raises NotImplementedError --
    there is no silent CPU path in this package.  Pass ``flood="third-party-cpu"`` to run, explicitly, the same
    scipy / scikit-image call the reference makes on the GPU-made cost image."""
    if flood != "third-party-cpu":
        raise NotImplementedError(
            "the watershed marker flood has no HIP implementation yet (bit-exact parallel tie-breaking is an open "
            "problem, DESIGN.md section 6); pass flood='third-party-cpu' to use the reference's own scipy/skimage call")
    from scipy import ndimage

    mask = np.memmap(tfile, shape=shape, dtype="uint8", mode="r+")
    if algorithm == "Watershed":
        try:
            from skimage.segmentation import watershed
        except ImportError as e:  # scikit-image is not part of this environment
            raise RuntimeError("algorithm 'Watershed' needs scikit-image for the flood step") from e
        tmp_image = cost_image(np.asarray(image), use_ww_wl, wl, ww, mg_size)
        tmp_mask = watershed(tmp_image, np.asarray(markers).astype("int16"), bstruct)
    else:
        tmp_image = cost_image(np.asarray(image), use_ww_wl, wl, ww, 0)
        mk = np.asarray(markers).astype("int16" if use_ww_wl else "int8")  # watershed_process.py:45,57
        tmp_mask = ndimage.watershed_ift(tmp_image, mk, bstruct)
    mask[:] = tmp_mask
    mask.flush()
    if q is not None:
        q.put(1)
