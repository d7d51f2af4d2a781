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


class IftDeviationWarning(UserWarning):
    """``do_watershed(..., algorithm="Watershed IFT")`` ran.  The reference's flood for that branch is
    ``scipy.ndimage.watershed_ift`` (watershed_process.py:44-46,54-57), whose C source has a linked-list defect
    (``ni_measure.c``: ``if (p->next || p->prev)`` misses the only element of a bucket) that pops some voxels late or
    never.  This package computes the DEFECT-FREE statement of that algorithm (oracle/ivx_oracle_wsz.c), so on realistic
    volumes a small share of the labels differs from the reference's: 87 372 of 2^27 voxels on bench.py's 512^3 volume,
    206 606 of 2^30 at 1024^3 (small fixtures such as the reference's own 5^3 test are identical).
    Emitted once per process, after the labels are written and ``q`` is signalled."""


_IFT_REFERENCE = "scipy.ndimage.watershed_ift"
_IFT_NOTE = ("defect-free statement of scipy's NI_WatershedIFT; live scipy deviates from its own algorithm through a linked-list "
             "defect in ni_measure.c, so labels can differ from the reference's on realistic volumes (87 372 / 2^27 voxels on "
             "bench.py's 512^3 volume, 206 606 / 2^30 at 1024^3; DESIGN.md section 6)")
_ift_warned = False

_MK_CODES = {np.dtype(np.uint8): L.U8, np.dtype(np.int8): L.I8, np.dtype(np.int16): L.I16, np.dtype(np.uint16): L.U16,
             np.dtype(np.int32): L.I32, np.dtype(np.int64): L.I64, np.dtype(np.bool_): L.U8}


def _box_view(a: np.ndarray) -> bool:
    """contiguous rows and non-negative pitches: what the strided host <-> device staging moves without a host-side pass"""
    isz = a.itemsize
    return a.strides[2] == isz and a.strides[1] >= a.shape[2] * isz and a.strides[0] >= a.strides[1] * a.shape[1]


def do_watershed(image, markers, tfile, shape, bstruct, algorithm, mg_size, use_ww_wl, wl, ww, q=None):
    """Same signature and side effects as watershed_process.do_watershed (:19-60): writes the uint8 label volume to
    the memmap `tfile` and signals ``q.put(1)``.  Cost image and flood run on the GPU; the markers go up in the caller's
    dtype and are cast (``astype("int16" | "int8")``, :39,45,52,57) on the device, the labels are downloaded straight into
    the memmap's pages (no ``tmp_mask``, no second host pass).  ``do_watershed.last_stats`` describes the run:

    * ``algorithm == "Watershed"``: the raster-tie variant of scikit-image's flood (see `watershed`) -- the number of
      adjacent tied markers of different labels, and a `MarkerTieWarning` when it is not 0;
    * ``algorithm == "Watershed IFT"``: ``{"reference": "scipy.ndimage.watershed_ift", "exact": False, "note": ...}`` and,
      once per process, an `IftDeviationWarning` -- the labels are those of scipy's documented algorithm, not of its
      defective linked list.

    Either warning comes after ``q.put(1)``."""
    global _ift_warned
    import os
    import time
    t_host = [time.perf_counter()] if os.environ.get("IVX_HOST_TIMING") else None

    def _mark(what):
        if t_host is not None:
            t = time.perf_counter()
            import sys
            print("py  do_watershed: %-28s %8.2f ms" % (what, (t - t_host[0]) * 1e3), file=sys.stderr)
            t_host[0] = t
    mask = np.memmap(tfile, shape=shape, dtype="uint8", mode="r+")
    image = np.asarray(image)
    if image.dtype != np.int16 or image.ndim not in (2, 3):
        raise TypeError("image must be a 2-D or 3-D int16 array")
    sk = algorithm == "Watershed"
    # watershed_process.py:39,45,52,57: int16 markers, except for the IFT flood of the min-shifted image (int8)
    mdt = np.dtype("int16" if (sk or use_ww_wl) else "int8")
    mk = np.asarray(markers)
    if mk.shape != image.shape:
        raise RuntimeError("input and markers must have equal shape")
    img3 = image if image.ndim == 3 else image[np.newaxis]
    mk3 = mk if mk.ndim == 3 else mk[np.newaxis]
    if mk3.dtype not in _MK_CODES or not _box_view(mk3):  # floats, exotic views: the host cast the reference does
        mk3 = np.ascontiguousarray(mk3.astype(mdt))
    gs = None
    if sk:  # int -> the same size on every axis (scipy semantics); tuple -> per axis
        sz = tuple(int(v) for v in mg_size) if np.ndim(mg_size) else (int(mg_size),) * image.ndim
        if len(sz) != image.ndim:
            raise RuntimeError("size must have one entry per image axis")
        gs = (ctypes.c_int * 3)(*((1,) * (3 - len(sz)) + sz))
    # (The library validates, uploads and floods first and writes into `dst` only when the flood has succeeded -- a rejected
    # marker array or an allocation failure leaves the caller's file untouched, as the reference's `mask[:] = tmp_mask` after the
    # flood does; only an error during the final download itself could leave the file partly written.)
    # `mask[:] = tmp_mask` (:58): the labels land in the memmap itself when it has the image's shape (it always does in the
    # reference's callers, styles.py:2102-2134); a broadcasting assignment keeps numpy's semantics through a temporary
    direct = mask.size == img3.size and tuple(d for d in mask.shape if d != 1) == tuple(d for d in img3.shape if d != 1)
    dst = mask.reshape(img3.shape) if direct else np.empty(img3.shape, np.uint8)
    stats = (ctypes.c_int64 * 16)()
    _mark("memmap + argument checks")
    L.check(L.lib().ivx_do_watershed_into(L.ptr(img3), L.i64(img3.shape), L.i64(img3.strides), _MK_CODES[mk3.dtype], L.ptr(mk3),
                                          L.i64(mk3.strides), L.I16 if mdt == np.int16 else L.I8, L.ptr(_strct27(bstruct, image.ndim)),
                                          int(sk), gs, int(bool(use_ww_wl)), ctypes.c_double(float(ww)), ctypes.c_double(float(wl)),
                                          L.ptr(dst), L.i64(dst.strides), stats), "do_watershed")
    if sk:
        do_watershed.last_stats = {"algorithm": algorithm, "reference": "skimage.segmentation.watershed",
                                   "tied_markers_of_different_labels": int(stats[6]), "exact": int(stats[6]) == 0}
    else:
        do_watershed.last_stats = {"algorithm": algorithm, "reference": _IFT_REFERENCE, "exact": False, "note": _IFT_NOTE,
                                   "tied_markers_of_different_labels": 0}
    _mark("ivx_do_watershed_into")
    if not direct:
        mask[:] = dst.reshape(image.shape)
    t_flush = time.perf_counter()
    mask.flush()  # (watershed_process.py:59: an msync of the label volume -- the reference's own cost, 20 - 35 ms at 512^3 on a disk-backed file)
    do_watershed.last_flush_ms = (time.perf_counter() - t_flush) * 1e3
    _mark("flush (msync)")
    if q is not None:
        q.put(1)
    # the warnings come LAST: with warnings promoted to errors (-W error) the labels are written and the caller waiting on
    # `q` (styles.py:2116-2134) has its signal before anything can raise
    if sk:
        _warn_ties(int(stats[6]), "do_watershed")
    elif not _ift_warned:
        _ift_warned = True
        warnings.warn("do_watershed(algorithm='Watershed IFT'): " + _IFT_NOTE, IftDeviationWarning, stacklevel=2)


do_watershed.last_stats = None
do_watershed.last_flush_ms = None
