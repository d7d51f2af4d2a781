"""Threshold and slab-projection entry points of ``invesalius.data.slice_.Slice`` on the GPU.

The reference methods live on the ``Slice`` singleton and read GUI state; the compute part is restated here as
free functions with the state passed explicitly (mask matrix, image matrix, threshold range), same argument
meaning, same in-place effects on the caller-owned ``(dz+1, dy+1, dx+1)`` uint8 mask matrix
(invesalius/data/mask.py:422-431) including the per-slice flag cells ``matrix[n, 0, 0]``.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from . import _lib as L

PROJECTION_NORMAL, PROJECTION_MaxIP, PROJECTION_MinIP, PROJECTION_MeanIP = 0, 1, 2, 3  # invesalius/constants.py:803-806


def _int_bounds(threshold_range):
    """numpy compares int16 voxels with python numbers; for integer voxels `v >= lo` == `v >= ceil(lo)`."""
    lo, hi = threshold_range
    lo = int(math.ceil(lo))
    hi = int(math.floor(hi))
    lo = max(min(lo, 2 ** 31 - 1), -(2 ** 31))
    hi = max(min(hi, 2 ** 31 - 1), -(2 ** 31))
    return lo, hi


def _threshold(mask_matrix, target_matrix, threshold_range, preserve, honour_flags):
    if mask_matrix.dtype != np.uint8 or mask_matrix.ndim != 3:
        raise TypeError("mask matrix must be a 3-D uint8 array")
    if target_matrix.dtype != np.int16 or target_matrix.ndim != 3:
        raise TypeError("image matrix must be a 3-D int16 array")
    if tuple(mask_matrix.shape) != tuple(s + 1 for s in target_matrix.shape):
        raise ValueError("mask matrix must be image shape + 1 per axis (invesalius/data/mask.py:422-431)")
    lo, hi = _int_bounds(threshold_range)
    L.check(L.lib().ivx_threshold_all_slices(
        L.ptr(target_matrix), L.i64(target_matrix.shape), L.i64(target_matrix.strides), ctypes.c_int(lo),
        ctypes.c_int(hi), ctypes.c_int(int(preserve)), ctypes.c_int(int(honour_flags)), L.ptr(mask_matrix),
        L.i64(mask_matrix.strides)), "do_threshold_to_all_slices")
    if hasattr(mask_matrix, "flush"):
        mask_matrix.flush()  # slice_.py:1769


def do_threshold_to_all_slices(mask_matrix, target_matrix, threshold_range):
    """Slice.do_threshold_to_all_slices (invesalius/data/slice_.py:1739-1769) with do_threshold_to_a_slice
    (:1722-1737) fused: slices whose flag ``mask_matrix[n,0,0]`` is 0 get ``255*in_range`` with existing
    1/2/253/254 preserved, and their flag set to 1; flagged slices are left untouched."""
    _threshold(mask_matrix, target_matrix, threshold_range, True, True)


def set_mask_threshold(mask_matrix, image_matrix, threshold_range):
    """Whole-volume branch of Slice.SetMaskThreshold (invesalius/data/slice_.py:1240-1247): no preserve rule,
    every slice written, every flag forced to 1."""
    _threshold(mask_matrix, image_matrix, threshold_range, False, False)


def do_threshold_to_a_slice(slice_matrix: np.ndarray, mask: np.ndarray, threshold) -> np.ndarray:
    """Slice.do_threshold_to_a_slice (invesalius/data/slice_.py:1722-1737): returns a NEW uint8 array with
    255*in_range and the existing 1/2/253/254 of `mask` preserved.  2-D (or any-D) int16 slice, same-shape uint8 mask."""
    if slice_matrix.dtype != np.int16 or mask.dtype != np.uint8 or slice_matrix.shape != mask.shape:
        raise TypeError("slice must be int16 and mask a uint8 array of the same shape")
    out = np.array(mask, dtype=np.uint8, copy=True)
    img3 = slice_matrix.reshape((1, -1, slice_matrix.shape[-1])) if slice_matrix.ndim != 3 else slice_matrix
    m3 = out.reshape(img3.shape)
    # full-matrix layout expected by the C entry point: one flag row / column in front of every axis
    big = np.zeros(tuple(s + 1 for s in img3.shape), np.uint8)
    big[1:, 1:, 1:] = m3
    _threshold(big, np.ascontiguousarray(img3), threshold, True, False)
    return big[1:, 1:, 1:].reshape(slice_matrix.shape).copy()


def set_mask_threshold_slice(slice_: np.ndarray, threshold_range) -> np.ndarray:
    """Per-slice preview branch of Slice.SetMaskThreshold (slice_.py:1253-1256):
    ``(255 * ((slice_ >= thresh_min) & (slice_ <= thresh_max))).astype("uint8")``."""
    return do_threshold_to_a_slice(slice_, np.zeros(slice_.shape, np.uint8), threshold_range)


def project(slab: np.ndarray, axis: int, projection: int) -> np.ndarray:
    """MaxIP / MinIP / MeanIP of a slab: ``np.array(tmp_array).max|min|mean(axis)``
    (invesalius/data/slice_.py:885-889, 969-973, 1056-1060)."""
    if slab.ndim != 3:
        raise TypeError("slab must be 3-D")
    code = L.dtype_code(slab, (L.U8, L.I16, L.U16))
    op = {PROJECTION_MaxIP: L.MIP_MAX, PROJECTION_MinIP: L.MIP_MIN, PROJECTION_MeanIP: L.MIP_MEAN}[projection]
    oshape = tuple(s for i, s in enumerate(slab.shape) if i != axis)
    out = np.empty(oshape, np.float64 if op == L.MIP_MEAN else slab.dtype)
    L.check(L.lib().ivx_mip_reduce(code, L.ptr(slab), L.i64(slab.shape), L.i64(slab.strides), int(axis), int(op),
                                   L.ptr(out), L.i64(out.strides)), "project")
    return out


PROJECTION_LMIP, PROJECTION_MIDA, PROJECTION_CONTOUR_MIP, PROJECTION_CONTOUR_LMIP, PROJECTION_CONTOUR_MIDA = 4, 5, 6, 7, 8
_AXIS = {"AXIAL": 0, "CORONAL": 1, "SAGITAL": 2}


def view_matrix(q_orientation, center) -> np.ndarray:
    """The 4x4 matrix ``Slice.get_image_slice`` builds for a reoriented view (invesalius/data/slice_.py:848-858):
    ``T1 . R^T . T0`` with ``T0 = translation(-cz, -cy, -cx)``, ``R = quaternion_matrix(q_orientation)`` (w, x, y, z;
    transformations.py:1261-1290, unit-normalised there, identity below 4 eps) and ``T1 = translation(cz, cy, cx)``;
    `center` is the reference's ``(cx, cy, cz)``.  C-contiguous float64, as apply_view_matrix_transform requires."""
    q = np.array(q_orientation, dtype=np.float64, copy=True)
    n = float(np.dot(q, q))
    R = np.identity(4)
    if n >= np.finfo(float).eps * 4.0:
        q *= np.sqrt(2.0 / n)
        o = np.outer(q, q)
        R = np.array([[1.0 - o[2, 2] - o[3, 3], o[1, 2] - o[3, 0], o[1, 3] + o[2, 0], 0.0],
                      [o[1, 2] + o[3, 0], 1.0 - o[1, 1] - o[3, 3], o[2, 3] - o[1, 0], 0.0],
                      [o[1, 3] - o[2, 0], o[2, 3] + o[1, 0], 1.0 - o[1, 1] - o[2, 2], 0.0],
                      [0.0, 0.0, 0.0, 1.0]])
    cx, cy, cz = (float(v) for v in center)
    T0, T1 = np.identity(4), np.identity(4)
    T0[:3, 3] = (-cz, -cy, -cx)
    T1[:3, 3] = (cz, cy, cx)
    M = np.identity(4)
    for m in (T1, R.T, T0):  # transformations.concatenate_matrices: left to right
        M = np.dot(M, m)
    return np.ascontiguousarray(M)


def get_image_slice(matrix: np.ndarray, orientation: str, slice_number: int, number_slices: int = 1, inverted: bool = False,
                    border_size: float = 1.0, type_projection: int = PROJECTION_NORMAL, window_level=0, q_orientation=None,
                    center=None, spacing=None, interp_method: int = 2) -> np.ndarray:
    """The array work of ``Slice.get_image_slice`` (invesalius/data/slice_.py:832-1119): the slab
    ``matrix[n : n + number_slices]`` along the view axis (one slice for PROJECTION_NORMAL); when the view is reoriented
    (``np.any(q_orientation[1:])``, :847,862,948,1035) the slab is resampled from the whole volume through
    `view_matrix(q_orientation, center)` with ``apply_view_matrix_transform(matrix, spacing, M, n, orientation,
    interp_method, matrix.min(), slab)`` (HIP: csrc/k_transform.hip) before anything else; then reversed when `inverted`,
    then the projection -- max / min / mean, MIDA, the three contour MIPs -- with the reference's own arguments, its quirk
    included (the window LEVEL goes in for level and width alike: :898-900,906-945).  LMIP raises AttributeError like the
    reference (`mips.lmip` is commented out of invesalius_rs/__init__.py:83).  The buffer_slices cache stays with the
    caller; `interp_method` defaults to the reference's (Slice.interp_method = 2, slice_.py:119)."""
    from . import invesalius_rs as mips

    ax = _AXIS[orientation]
    if type_projection == PROJECTION_NORMAL:
        number_slices = 1
    sl = [slice(None)] * 3
    sl[ax] = slice(slice_number, slice_number + number_slices)
    tmp = np.array(matrix[tuple(sl)])
    if q_orientation is not None and np.any(np.asarray(q_orientation)[1:]):
        if center is None or spacing is None:
            raise TypeError("a reoriented view needs `center` (cx, cy, cz) and `spacing`")
        mips.apply_view_matrix_transform(matrix, spacing, view_matrix(q_orientation, center), slice_number, orientation,
                                         interp_method, matrix.min(), tmp)
    oshape = tuple(s for i, s in enumerate(tmp.shape) if i != ax)
    if type_projection == PROJECTION_NORMAL:
        return tmp.reshape(oshape)
    if inverted:
        rev = [slice(None)] * 3
        rev[ax] = slice(None, None, -1)
        tmp = tmp[tuple(rev)]
    if type_projection in (PROJECTION_MaxIP, PROJECTION_MinIP, PROJECTION_MeanIP):
        return project(tmp, ax, type_projection)
    if type_projection == PROJECTION_LMIP:
        raise AttributeError("module 'invesalius_rs' has no attribute 'lmip'")
    out = np.empty(oshape, tmp.dtype)
    if type_projection == PROJECTION_MIDA:
        mips.mida(tmp, ax, int(window_level), int(window_level), out)
    elif type_projection in (PROJECTION_CONTOUR_MIP, PROJECTION_CONTOUR_LMIP, PROJECTION_CONTOUR_MIDA):
        mips.fast_countour_mip(tmp, border_size, ax, window_level, window_level, type_projection - PROJECTION_CONTOUR_MIP, out)
    else:  # (:946-947: anything else shows the plain slice)
        sl[ax] = slice_number
        return np.array(matrix[tuple(sl)])
    return out


def apply_reorientation(matrix: np.ndarray, spacing, q_orientation, center, interp_method: int = 2, masks=()):
    """The array work of ``Slice.apply_reorientation`` (invesalius/data/slice_.py:1969-2068): bake the view's rotation into
    the data.  `matrix` (the int16 image, e.g. the ``matrix.dat`` memmap) is resampled IN PLACE from a copy of itself
    through ``M = view_matrix(q_orientation, center)`` -- ``apply_view_matrix_transform(copy, spacing, M, 0, "AXIAL",
    interp_method, copy.min(), matrix)`` (:1979-1988; HIP: csrc/k_transform.hip) -- and flushed.  Then every mask of `masks`
    (objects with ``.matrix``, the padded ``(dz+1, dy+1, dx+1)`` uint8 array, and ``.was_edited``): an edited mask goes
    through the SAME call with nearest-neighbour interpolation and cval 0 (:2034-2043) -- on the whole padded matrix, flag
    planes included, exactly as the reference does (its voxel (z, y, x) sits at matrix index (z+1, y+1, x+1), so the mask
    turns about a point one voxel off the image's: a reference quirk, kept) -- a threshold mask is cleared (:2055-2062);
    ``mask.clear_history()`` is called where it exists.  Returns the view state the reference leaves behind:
    ``(q_orientation, center) = ((1, 0, 0, 0), [s * d / 2 for d, s in zip(shape[::-1], spacing)])`` (:2002-2003)."""
    from . import invesalius_rs as mips

    if matrix.ndim != 3:
        raise TypeError("matrix must be a 3-D array")
    M = view_matrix(q_orientation, center)
    mcopy = np.array(matrix)  # (the reference's temporary memmap copy)
    mips.apply_view_matrix_transform(mcopy, spacing, M, 0, "AXIAL", int(interp_method), mcopy.min(), matrix)
    del mcopy
    if hasattr(matrix, "flush"):
        matrix.flush()
    new_mask_shape = tuple(s + 1 for s in matrix.shape)
    for mask in masks:
        mm = mask.matrix
        if tuple(mm.shape) != new_mask_shape:  # (:2028-2030 would recreate it; the image's shape never changes here)
            raise ValueError("mask matrix must be image shape + 1 per axis (invesalius/data/mask.py:422-431)")
        if getattr(mask, "was_edited", False):
            mask_copy = np.array(mm)
            mips.apply_view_matrix_transform(mask_copy, spacing, M, 0, "AXIAL", 0, 0, mm)
            del mask_copy
        else:
            mm[:] = 0
        if hasattr(mm, "flush"):
            mm.flush()
        if hasattr(mask, "clear_history"):
            mask.clear_history()
    return np.array((1, 0, 0, 0)), [(s * d / 2.0) for (d, s) in zip(matrix.shape[::-1], spacing)]


def calc_image_area(mask_matrix: np.ndarray, spacing) -> float:
    """Slice.calc_image_area (invesalius/data/slice_.py:2296-2322) after its threshold step: the exposed-face area of
    ``mask_matrix[1:,1:,1:] > 127`` for ``spacing = (sx, sy, sz)``.  Computed from the uint8 mask on the GPU (the
    reference builds ``bin_img * 1.0`` and calls convolve_non_zero(..., cval=1).sum()); per-voxel terms are identical,
    the final sum is a tree sum instead of numpy's pairwise one (differs by rounding only)."""
    if mask_matrix.dtype != np.uint8 or mask_matrix.ndim != 3:
        raise TypeError("mask matrix must be a 3-D uint8 array")
    inner = mask_matrix[1:, 1:, 1:]
    sp = (ctypes.c_double * 3)(*[float(v) for v in spacing])
    area = ctypes.c_double(0.0)
    L.check(L.lib().ivx_mask_area(L.ptr(inner), L.i64(inner.shape), L.i64(inner.strides), sp, ctypes.byref(area)), "mask_area")
    return float(area.value)


BOOLEAN_UNION, BOOLEAN_DIFF, BOOLEAN_AND, BOOLEAN_XOR = 1, 2, 3, 4  # invesalius/constants.py:818-821


def do_boolean_op(op: int, m1_matrix: np.ndarray, m2_matrix: np.ndarray) -> np.ndarray:
    """Slice.do_boolean_op (slice_.py:1878-1923), the array part: a new padded mask matrix filled with 1 (flags
    "thresholded"), interior = 255 where op(m1 > 2, m2 > 2).  Both inputs are padded (dz+1, dy+1, dx+1) matrices that
    the caller has brought up to date (the reference calls do_threshold_to_all_slices on both first)."""
    if m1_matrix.shape != m2_matrix.shape or m1_matrix.dtype != np.uint8 or m2_matrix.dtype != np.uint8:
        raise TypeError("two uint8 mask matrices of the same shape")
    if op not in (BOOLEAN_UNION, BOOLEAN_DIFF, BOOLEAN_AND, BOOLEAN_XOR):
        raise ValueError("unknown boolean operation %r" % (op,))
    out = np.ones(m1_matrix.shape, np.uint8)
    a, b, m = m1_matrix[1:, 1:, 1:], m2_matrix[1:, 1:, 1:], out[1:, 1:, 1:]
    L.check(L.lib().ivx_mask_boolean(int(op), L.ptr(a), L.i64(a.strides), L.ptr(b), L.i64(b.strides), L.ptr(m),
                                     L.i64(m.strides), L.i64(m.shape)), "mask_boolean")
    return out


def calc_image_density(image: np.ndarray, mask_matrix: np.ndarray):
    """Slice.calc_image_density (slice_.py:2284-2297): (min, max, mean, std) of ``image[mask[1:,1:,1:] > 127]``, or
    four zeros when the mask selects nothing.  min / max are exact; mean and std come from exact integer sums, formed
    in float64 (numpy's two-pass std differs from sqrt(E[x^2] - mean^2) by rounding only)."""
    if image.dtype != np.int16 or image.ndim != 3:
        raise TypeError("image must be a 3-D int16 array")
    inner = mask_matrix[1:, 1:, 1:]
    if inner.shape != image.shape or mask_matrix.dtype != np.uint8:
        raise TypeError("mask matrix must be uint8 and one voxel larger than the image on every axis")
    out = np.zeros(5, np.float64)
    L.check(L.lib().ivx_masked_density_i16(L.ptr(image), L.i64(image.strides), L.ptr(inner), L.i64(inner.strides),
                                           L.i64(image.shape), L.ptr(out)), "masked_density")
    cnt, s1, s2, lo, hi = out
    if cnt == 0:
        return 0, 0, 0, 0
    mean = s1 / cnt
    var = max(s2 / cnt - mean * mean, 0.0)
    return int(lo), int(hi), float(mean), float(np.sqrt(var))
