"""``invesalius.data.watershed_process`` on the GPU -- the deterministic stages.

``do_watershed`` (watershed_process.py:19-60) is: [window/level LUT | subtract the minimum] -> uint16 ->
[morphological gradient] -> marker flood (``skimage.segmentation.watershed`` or ``scipy.ndimage.watershed_ift``)
-> uint8 labels, followed in the caller by the merge rule (styles.py:2147-2152).

Everything runs in HIP kernels: the LUT / min-shift, the 3x3x3 (or larger, odd) morphological gradient, both marker
floods -- ``watershed_ift`` (csrc/k_wsift.hip, the zone formulation of scipy's serial bucket flood) and ``watershed``
(csrc/k_wssk.hip, the run formulation of scikit-image's serial (value, age) heap flood) -- and the merge.  There is no
CPU path in this package.
"""
from __future__ import annotations

import ctypes

import numpy as np

import warnings

from . import _lib as L


class MarkerTieWarning(UserWarning):
    """The scikit-image branch met adjacent marker voxels of DIFFERENT labels and EQUAL image value.  scikit-image pops
    such markers in an order that depends on its heap's array layout (every push and pop before them), which no parallel
    flood can reproduce; this package takes them in raster order.  Where their basins meet before the cost decides, the
    labels may differ from scikit-image's.  Zero such pairs = identical to scikit-image by construction."""


def _warn_ties(n: int, where: str):
    if n > 0:
        warnings.warn("%s: %d adjacent tied markers of different labels -- taken in raster order, scikit-image's order "
                      "depends on its heap layout; labels can differ where those basins meet (DESIGN.md 6b)" % (where, n),
                      MarkerTieWarning, stacklevel=3)


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


def _strct27(strct, ndim):
    s = np.asarray(strct)
    if s.ndim != ndim or any(d != 3 for d in s.shape):
        raise RuntimeError("structure and input must have equal rank")  # scipy's message
    s3 = np.zeros((3, 3, 3), np.uint8)
    if ndim == 3:
        s3[:] = s.astype(bool)
    else:
        s3[1] = s.astype(bool)
    return s3


def watershed_ift(image: np.ndarray, markers: np.ndarray, structure=None, want_cost=False, want_stats=False):
    """``scipy.ndimage.watershed_ift(input, markers, structure)`` on the GPU (2-D or 3-D uint8 / uint16 input, int8 /
    int16 markers >= 0), labels returned in the markers' dtype.  Same argument checks as scipy: TypeError for other
    input dtypes, RuntimeError for shape mismatches."""
    image = np.asarray(image)
    markers = np.asarray(markers)
    if image.dtype.type not in (np.uint8, np.uint16):
        raise TypeError("only 8 and 16 unsigned inputs are supported")
    if markers.dtype.type not in (np.int8, np.int16):
        raise TypeError("markers must be int8 or int16 (the reference casts them: watershed_process.py:45,57)")
    if image.ndim not in (2, 3) or markers.shape != image.shape:
        raise RuntimeError("input and markers must have equal shape")
    if structure is None:
        from .mask import _structure
        structure = _structure(image.ndim, 1)
    s3 = _strct27(structure, image.ndim)
    img = np.ascontiguousarray(image)
    mk = np.ascontiguousarray(markers)
    shp = img.shape if img.ndim == 3 else (1,) + img.shape
    out = np.empty(mk.shape, mk.dtype)
    cost = np.empty(img.shape, np.uint16) if want_cost else None
    stats = (ctypes.c_int64 * 16)()
    L.check(L.lib().ivx_watershed_ift(L.U8 if img.dtype == np.uint8 else L.U16, L.ptr(img), L.i64(shp),
                                      L.I16 if mk.dtype == np.int16 else L.I8, L.ptr(mk), L.ptr(s3), L.ptr(out),
                                      L.ptr(cost) if want_cost else None, stats), "watershed_ift")
    res = (out,)
    if want_cost:
        res += (cost,)
    if want_stats:
        names = ("rounds", "tile_visits", "levels", "time_stamps", "markers", "entries", "tiles", "tile_sweeps", "us_costs", "us_zones",
                 "us_bucket", "us_levels", "us_labels", "cost_levels", "cost_level_rounds", "cost_level_voxels")
        res += ({k: int(v) for k, v in zip(names, stats) if not k.startswith("_")},)
    return res[0] if len(res) == 1 else res


def watershed(image: np.ndarray, markers: np.ndarray, connectivity=None, want_cost=False, want_stats=False):
    """``skimage.segmentation.watershed(image, markers, connectivity)`` on the GPU, for the call the reference makes
    (watershed_process.py:39,52; styles.py:1958,1975): `connectivity` is the 3x3(x3) structure array, no offset, mask,
    compactness or watershed lines.  2-D or 3-D uint8 / uint16 image, int8 / int16 markers; int32 labels like
    scikit-image.  This is the RASTER-TIE variant of that flood: ``stats["tied_markers_of_different_labels"] == 0``
    means identical to scikit-image by construction; otherwise a `MarkerTieWarning` is raised with the count, because
    scikit-image's own order among such markers depends on its heap layout."""
    image = np.asarray(image)
    markers = np.asarray(markers)
    if image.dtype.type not in (np.uint8, np.uint16):
        raise TypeError("image must be uint8 or uint16 (the reference passes the uint16 gradient image)")
    if markers.dtype.type not in (np.int8, np.int16):
        raise TypeError("markers must be int8 or int16 (the reference casts them: watershed_process.py:39,52)")
    if image.ndim not in (2, 3):
        raise ValueError("image must be 2-D or 3-D")
    if markers.shape != image.shape:  # scikit-image's message (_validate_inputs)
        raise ValueError("`markers` (shape {}) must have same shape as `image` (shape {})".format(markers.shape, image.shape))
    if connectivity is None:
        from .mask import _structure
        connectivity = _structure(image.ndim, 1)
    s3 = _strct27(connectivity, image.ndim)
    img = np.ascontiguousarray(image)
    mk = np.ascontiguousarray(markers)
    shp = img.shape if img.ndim == 3 else (1,) + img.shape
    out = np.zeros(mk.shape, np.int32)
    cost = np.empty(img.shape, np.uint16) if want_cost else None
    stats = (ctypes.c_int64 * 16)()
    L.check(L.lib().ivx_watershed_sk(L.U8 if img.dtype == np.uint8 else L.U16, L.ptr(img), L.i64(shp),
                                     L.I16 if mk.dtype == np.int16 else L.I8, L.ptr(mk), L.ptr(s3), L.ptr(out),
                                     L.ptr(cost) if want_cost else None, stats), "watershed")
    _warn_ties(int(stats[6]), "watershed")
    res = (out,)
    if want_cost:
        res += (cost,)
    if want_stats:
        names = ("rounds", "tile_visits", "levels", "generations", "markers", "generation0", "tied_markers_of_different_labels",
                 "frontier_launches", "us_costs", "us_generation0", "us_levels", "us_labels", "basin_rounds", "generation_steps", "small_level_runs", "tile_rounds")
        res += ({k: int(v) for k, v in zip(names, stats) if not k.startswith("_")},)
    return res[0] if len(res) == 1 else res


def do_watershed(image, markers, tfile, shape, bstruct, algorithm, mg_size, use_ww_wl, wl, ww, q=None):
    """Same signature and side effects as watershed_process.do_watershed (:19-60): writes the uint8 label volume to
    the memmap `tfile` and signals ``q.put(1)``.  Cost image and flood run on the GPU.  With ``algorithm ==
    "Watershed"`` this is the raster-tie variant of scikit-image's flood (see `watershed`): the number of adjacent tied
    markers of different labels is kept in ``do_watershed.last_stats`` and a `MarkerTieWarning` says so when it is not 0."""
    mask = np.memmap(tfile, shape=shape, dtype="uint8", mode="r+")
    image = np.asarray(image)
    if image.dtype != np.int16 or image.ndim not in (2, 3):
        raise TypeError("image must be a 2-D or 3-D int16 array")
    sk = algorithm == "Watershed"
    # watershed_process.py:39,45,52,57: int16 markers, except for the IFT flood of the min-shifted image (int8)
    mk = np.ascontiguousarray(np.asarray(markers).astype("int16" if (sk or use_ww_wl) else "int8"))
    if mk.shape != image.shape:
        raise RuntimeError("input and markers must have equal shape")
    img3 = image if image.ndim == 3 else image[np.newaxis]
    gs = None
    if sk:  # int -> the same size on every axis (scipy semantics); tuple -> per axis
        sz = tuple(int(v) for v in mg_size) if np.ndim(mg_size) else (int(mg_size),) * image.ndim
        if len(sz) != image.ndim:
            raise RuntimeError("size must have one entry per image axis")
        gs = (ctypes.c_int * 3)(*((1,) * (3 - len(sz)) + sz))
    tmp_mask = np.empty(img3.shape, np.uint8)
    stats = (ctypes.c_int64 * 16)()
    # one call: image and markers up, uint8 labels back (the cost / gradient image never leaves the device)
    L.check(L.lib().ivx_do_watershed(L.ptr(img3), L.i64(img3.shape), L.i64(img3.strides), L.I16 if mk.dtype == np.int16 else L.I8,
                                     L.ptr(mk), L.ptr(_strct27(bstruct, image.ndim)), int(sk), gs, int(bool(use_ww_wl)),
                                     ctypes.c_double(float(ww)), ctypes.c_double(float(wl)), L.ptr(tmp_mask), stats), "do_watershed")
    do_watershed.last_stats = {"algorithm": algorithm, "tied_markers_of_different_labels": int(stats[6]) if sk else 0}
    tmp_mask = tmp_mask.reshape(image.shape)
    mask[:] = tmp_mask
    mask.flush()
    if q is not None:
        q.put(1)
    # the warning comes LAST: with warnings promoted to errors (-W error) the labels are written and the caller waiting on
    # `q` (styles.py:2116-2134) has its signal before anything can raise
    if sk:
        _warn_ties(int(stats[6]), "do_watershed")


do_watershed.last_stats = None
