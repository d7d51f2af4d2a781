"""Drop-in for the hot-path names of the reference's native module ``invesalius_rs``
(invesalius_rs/__init__.py:11-111; PyO3 bindings invesalius_rs/src/lib.rs:22-66).

Same function names, argument order and in-place effects on caller-owned numpy arrays (strided views such as
``mask.matrix[1:, 1:, 1:]`` are accepted); the arithmetic runs in HIP kernels behind the C ABI of libivx.so.
Errors follow SURVEY.md 8(b): dtype mismatch -> TypeError, out-of-bounds seed -> IndexError (the reference panics
with pyo3_runtime.PanicException), NumCast failure -> ValueError.

Usage in the reference tree is unchanged:   ``from invesalius3_amd import invesalius_rs as floodfill``.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib as L


def _seeds(seeds) -> np.ndarray:
    tuple_seeds = [tuple(s) for s in seeds]  # invesalius_rs/__init__.py:29
    s = np.array(tuple_seeds, dtype=np.int64).reshape(-1, 3)
    return np.ascontiguousarray(s)


def _strct(strct) -> np.ndarray:
    s = np.ascontiguousarray(strct, dtype=np.uint8)  # invesalius_rs/__init__.py:31
    if s.ndim != 3:
        raise TypeError("strct must be 3-D")
    return s


def floodfill_threshold(data, seeds, t0, t1, fill, strct, out):
    """generic_floodfill_threshold (invesalius_rs/src/floodfill.rs:96-166) through the wrapper semantics of
    invesalius_rs/__init__.py:21-40: seeds are (x, y, z); for int16 images t0/t1/fill are truncated with int(); for
    uint8 images they must be integers in 0..255 (TypeError / OverflowError like the binding's u8 extraction); `out`
    (uint8, same shape) receives `fill` on the connected in-range region; voxels of `out` that already hold `fill` are
    barriers."""
    if data.ndim != 3 or out.ndim != 3 or tuple(data.shape) != tuple(out.shape):
        raise TypeError("data and out must be 3-D arrays of the same shape")
    if out.dtype != np.uint8:
        raise TypeError("out must be uint8")
    code = L.dtype_code(data, (L.U8, L.I16, L.F64))
    strct_u8 = _strct(strct)
    if data.dtype == np.int16:          # the wrapper's int() truncation (invesalius_rs/__init__.py:32-35) ...
        t0, t1, fill = _fits(int(t0), data.dtype, "t0"), _fits(int(t1), data.dtype, "t1"), _fits(int(fill), np.dtype(np.uint8), "fill")
    elif data.dtype == np.uint8:        # ... does not cover uint8: t0 / t1 / fill reach `extract::<u8>()` as they are
        for name, v in (("t0", t0), ("t1", t1), ("fill", fill)):
            if not isinstance(v, (int, np.integer)) or isinstance(v, (bool, np.bool_)):
                raise TypeError("%s must be an integer for uint8 data (the reference's u8 extraction rejects %r)" % (name, v))
        t0, t1, fill = (_fits(v, data.dtype, n) for n, v in (("t0", t0), ("t1", t1), ("fill", fill)))
    else:
        t0, t1, fill = float(t0), float(t1), float(fill)
        if not 0 <= fill <= 255 or fill != int(fill):
            raise OverflowError("fill=%r does not fit the uint8 out array" % (fill,))
    s = _seeds(seeds)
    L.check(L.lib().ivx_floodfill_threshold(
        code, L.ptr(data), L.i64(data.shape), L.i64(data.strides), L.ptr(s), ctypes.c_int64(len(s)),
        ctypes.c_double(t0), ctypes.c_double(t1), ctypes.c_int(int(fill) & 0xFF), L.ptr(strct_u8),
        L.i64(strct_u8.shape), L.ptr(out), L.i64(out.strides)), "floodfill_threshold")


def floodfill_threshold_inplace(data, seeds, t0, t1, fill, strct):
    """generic_floodfill_threshold_inplace (invesalius_rs/src/floodfill.rs:168-237; wrapper
    invesalius_rs/__init__.py:43-54): predicate and fill on the same array (mask relabelling)."""
    if data.ndim != 3:
        raise TypeError("data must be 3-D")
    code = L.dtype_code(data, (L.U8, L.I16, L.F64))
    strct_u8 = _strct(strct)
    s = _seeds(seeds)
    L.check(L.lib().ivx_floodfill_threshold_inplace(
        code, L.ptr(data), L.i64(data.shape), L.i64(data.strides), L.ptr(s), ctypes.c_int64(len(s)),
        ctypes.c_double(t0), ctypes.c_double(t1), ctypes.c_double(fill), L.ptr(strct_u8), L.i64(strct_u8.shape)),
        "floodfill_threshold_inplace")


def floodfill(data, i, j, k, v, fill, out):
    """floodfill (invesalius_rs/__init__.py:10 -> floodfill_py.rs:88-135 -> floodfill_internal floodfill.rs:5-49):
    the 6-neighbour component of ``data == v`` around the seed (i, j, k) = (x, y, z) receives `fill` in `out`; the seed
    itself is filled and expanded whatever its value, and voxels of `out` that already hold `fill` are barriers.
    dtype pairs of the binding: int16 / uint8 / float64 data with a uint8 `out`; `v` and `fill` must fit their dtype
    (``v.extract::<i16>()``, ``fill.extract::<u8>()``)."""
    if data.ndim != 3 or out.ndim != 3 or tuple(data.shape) != tuple(out.shape):
        raise TypeError("data and out must be 3-D arrays of the same shape")
    if out.dtype != np.uint8:
        raise TypeError("out must be uint8")
    code = L.dtype_code(data, (L.U8, L.I16, L.F64))
    if data.dtype.kind in "iu" and not isinstance(v, (int, np.integer)):
        raise TypeError("v must be an integer for %s data" % data.dtype)
    v = _fits(v, data.dtype, "v")
    fill = _fits(fill, np.dtype(np.uint8), "fill")
    L.check(L.lib().ivx_floodfill(code, L.ptr(data), L.i64(data.shape), L.i64(data.strides), ctypes.c_int64(int(i)),
                                  ctypes.c_int64(int(j)), ctypes.c_int64(int(k)), ctypes.c_double(v), ctypes.c_int(fill),
                                  L.ptr(out), L.i64(out.strides)), "floodfill")


def floodfill_auto_threshold(data, seeds, p, fill, out):
    """floodfill_auto_threshold (invesalius_rs/__init__.py:57-65 -> floodfill_py.rs:12-85): int16 data, uint8 out.
    Every seed (x, y, z) is filled and expanded; from a voxel of value v the flood steps to the 6-neighbours whose
    value lies in [ceil(v * (1 - p)), floor(v * (1 + p))] (float32 products, cast to int16 the way Rust's ``as`` does)
    and whose `out` byte is not `fill` yet.  The result does not depend on the order of the queue: it is the set of
    voxels reachable from the seeds along such steps."""
    if data.ndim != 3 or out.ndim != 3 or tuple(data.shape) != tuple(out.shape):
        raise TypeError("data and out must be 3-D arrays of the same shape")
    if data.dtype != np.int16 or out.dtype != np.uint8:
        raise TypeError("data must be int16 and out uint8")
    fill = _fits(fill, np.dtype(np.uint8), "fill")
    s = _seeds(seeds)
    L.check(L.lib().ivx_floodfill_auto_threshold(
        L.ptr(data), L.i64(data.shape), L.i64(data.strides), L.ptr(s), ctypes.c_int64(len(s)), ctypes.c_float(float(p)),
        ctypes.c_int(fill), L.ptr(out), L.i64(out.strides)), "floodfill_auto_threshold")


def jump_flooding(distance_map, map_owners, sites, normalize):
    """jump_flooding (invesalius_rs/__init__.py:76-80 -> floodfill_py.rs:262-275 -> jump_flooding_internal
    floodfill.rs:298-507): 3-D jump flooding.  `distance_map` (float32) and `map_owners` (int32; 1-based index into
    `sites`, 0 = no owner) are updated in place; `sites` is an (n, 3) int32 array of (z, y, x) rows; with `normalize`
    every site moves to the integer centroid of its cell and the distances are divided by the cell's maximum."""
    if distance_map.ndim != 3 or map_owners.ndim != 3 or tuple(distance_map.shape) != tuple(map_owners.shape):
        raise TypeError("distance_map and map_owners must be 3-D arrays of the same shape")
    if distance_map.dtype != np.float32 or map_owners.dtype != np.int32:
        raise TypeError("distance_map must be float32 and map_owners int32")
    sites = np.asarray(sites)
    if sites.dtype != np.int32 or sites.ndim != 2:
        raise TypeError("sites must be a 2-D int32 array")
    if sites.shape[0] and sites.shape[1] < 3:
        raise IndexError("sites rows need (z, y, x)")  # the Rust code indexes columns 0..2
    s = np.ascontiguousarray(sites[:, :3])
    L.check(L.lib().ivx_jump_flooding(L.ptr(distance_map), L.i64(distance_map.strides), L.ptr(map_owners),
                                      L.i64(map_owners.strides), L.i64(distance_map.shape), L.ptr(s),
                                      ctypes.c_int64(len(s)), ctypes.c_int(1 if normalize else 0)), "jump_flooding")


def _proj_out_shape(image, axis):
    a = 2 if axis not in (0, 1, 2) else axis
    return tuple(d for i, d in enumerate(image.shape) if i != a)


def _fits(v, dtype, name):
    """`wl.extract::<i16>()` / `::<u8>()` (mips_py.rs:172-184): a Python int outside the image dtype is an
    OverflowError in the reference."""
    if dtype.kind in "iu":
        info = np.iinfo(dtype)
        if not info.min <= int(v) <= info.max:
            raise OverflowError("%s=%r out of range for %s" % (name, v, dtype))
        return int(v)
    return float(v)


def mida(image: np.ndarray, axis: int, wl: int, ww: int, out: np.ndarray):
    """mida (invesalius_rs/__init__.py:91-95 -> mips_py.rs:161-202 -> mida_internal mips.rs:102-168).
    dtype pairs: int16->int16, uint8->uint8, float64->uint8."""
    if image.ndim != 3 or out.ndim != 2:
        raise TypeError("Invalid image or output type")
    pairs = {(np.dtype(np.int16), np.dtype(np.int16)), (np.dtype(np.uint8), np.dtype(np.uint8)),
             (np.dtype(np.float64), np.dtype(np.uint8))}
    if (image.dtype, out.dtype) not in pairs:
        raise TypeError("Invalid image or output type")
    if tuple(out.shape) != _proj_out_shape(image, axis):
        raise ValueError("out has the wrong shape")
    wl, ww = _fits(int(wl), image.dtype, "wl"), _fits(int(ww), image.dtype, "ww")
    L.check(L.lib().ivx_mida(L.DT[image.dtype], L.ptr(image), L.i64(image.shape), L.i64(image.strides), int(axis),
                             ctypes.c_double(wl), ctypes.c_double(ww), L.DT[out.dtype], L.ptr(out),
                             L.i64(out.strides)), "mida")


def lmip(image: np.ndarray, axis: int, tmin, tmax, out: np.ndarray):
    """lmip (invesalius_rs/src/mips.rs:7-86).  The reference calls it (slice_.py:892,980,1063) but never exports it
    (invesalius_rs/__init__.py:83 is commented out -> AttributeError); exported here."""
    if image.ndim != 3 or out.ndim != 2 or image.dtype != out.dtype:
        raise TypeError("Invalid image or output type")
    code = L.dtype_code(image, (L.U8, L.I16, L.F64))
    if axis in (0, 1, 2) and tuple(out.shape) != _proj_out_shape(image, axis):
        raise ValueError("out has the wrong shape")
    tmin, tmax = _fits(tmin, image.dtype, "tmin"), _fits(tmax, image.dtype, "tmax")
    L.check(L.lib().ivx_lmip(code, L.ptr(image), L.i64(image.shape), L.i64(image.strides), int(axis),
                             ctypes.c_double(tmin), ctypes.c_double(tmax), L.ptr(out), L.i64(out.strides)), "lmip")


def fast_countour_mip(image: np.ndarray, n: float, axis: int, wl: int, ww: int, tmip: int, out: np.ndarray):
    """fast_countour_mip (invesalius_rs/__init__.py:98-101 -> mips_py.rs:204-253 -> mips.rs:215-279).
    tmip 0 = MIP, 1 = LMIP(700, 3033), 2 = MIDA of the contour volume; out dtype == image dtype."""
    if image.ndim != 3 or out.ndim != 2 or image.dtype != out.dtype:
        raise TypeError("Invalid image or output type")
    code = L.dtype_code(image, (L.U8, L.I16, L.F64))
    if tuple(out.shape) != _proj_out_shape(image, axis):
        raise ValueError("out has the wrong shape")
    wl, ww = _fits(int(wl), image.dtype, "wl"), _fits(int(ww), image.dtype, "ww")
    L.check(L.lib().ivx_fast_countour_mip(code, L.ptr(image), L.i64(image.shape), L.i64(image.strides),
                                          ctypes.c_float(n), int(axis), ctypes.c_double(wl), ctypes.c_double(ww),
                                          int(tmip), L.ptr(out), L.i64(out.strides)), "fast_countour_mip")


def fill_holes_automatically(mask: np.ndarray, labels: np.ndarray, nlabels: int, max_size: int) -> bool:
    """fill_holes_automatically (floodfill_py.rs:233-249 -> floodfill.rs:51-94): voxels whose uint32 label has
    0 < size <= max_size become 254 in `mask` (in place); returns whether anything was modified."""
    if mask.dtype != np.uint8 or labels.dtype != np.uint32:
        raise TypeError("mask must be uint8 and labels uint32")
    if mask.ndim != 3 or tuple(mask.shape) != tuple(labels.shape):
        raise TypeError("mask and labels must be 3-D arrays of the same shape")
    modified = ctypes.c_int(0)
    L.check(L.lib().ivx_fill_holes_automatically(
        L.ptr(mask), L.i64(mask.shape), L.i64(mask.strides), L.ptr(labels), L.i64(labels.strides),
        ctypes.c_uint32(int(nlabels)), ctypes.c_uint32(int(max_size)), ctypes.byref(modified)),
        "fill_holes_automatically")
    return bool(modified.value)


_ORIENTATION = {"AXIAL": 0, "CORONAL": 1, "SAGITAL": 2}


def apply_view_matrix_transform(volume, spacing, m, n, orientation, minterpol, cval, out):
    """apply_view_matrix_transform (invesalius_rs/src/transforms_py.rs:95-147 -> transforms.rs:9-55): resample `volume`
    through the 4x4 matrix `m` into `out` (same dtype: int16 / uint8 / float64).  `m` must be a C-contiguous float64
    4x4 (the reference calls ``m.as_slice().unwrap()``); `cval` must fit the dtype (``cval.extract::<T>()``)."""
    if volume.ndim != 3 or out.ndim != 3 or volume.dtype != out.dtype:
        raise TypeError("Invalid volume or output type")
    code = L.dtype_code(volume, (L.U8, L.I16, L.F64))
    mm = np.asarray(m)
    if mm.dtype != np.float64 or mm.shape != (4, 4) or not mm.flags["C_CONTIGUOUS"]:
        raise TypeError("m must be a C-contiguous float64 4x4 matrix")
    sp = np.ascontiguousarray(spacing, dtype=np.float64)
    if sp.shape != (3,):
        raise TypeError("spacing must have 3 entries")
    cval = _fits(cval, volume.dtype, "cval")
    L.check(L.lib().ivx_apply_view_matrix_transform(
        code, L.ptr(volume), L.i64(volume.shape), L.i64(volume.strides), L.ptr(sp), L.ptr(mm), ctypes.c_int64(int(n)),
        _ORIENTATION.get(orientation, -1), int(minterpol), ctypes.c_double(cval), L.ptr(out), L.i64(out.shape),
        L.i64(out.strides)), "apply_view_matrix_transform")


# ---------------------------------------------------------------------------------------------------------------------
# context-aware smoothing: invesalius_rs/__init__.py:114-275 (Mesh, ca_smoothing) over mesh_py.rs context_aware_smoothing
# ---------------------------------------------------------------------------------------------------------------------
_FACE_DTYPES = (np.int64, np.int32, np.uint64, np.uint32)


def context_aware_smoothing(vertices, faces, normals, t, tmax, bmin, n_iters):
    """mesh_py.rs:7-330.  ``vertices`` (N,3) float32/float64 is smoothed IN PLACE; ``faces`` is the vtkCellArray
    layout the reference passes, (M,4) rows ``[3, v0, v1, v2]`` of int64/int32/uint64/uint32; ``normals`` (M,3)
    float32/float64 cell normals.  Any other dtype is the TypeError PyO3 raises for an unmatched enum."""
    if not isinstance(vertices, np.ndarray) or vertices.dtype not in (np.float32, np.float64):
        raise TypeError("vertices must be a float32 or float64 ndarray")
    if not isinstance(faces, np.ndarray) or faces.dtype not in _FACE_DTYPES:
        raise TypeError("faces must be an int64, int32, uint64 or uint32 ndarray")
    if not isinstance(normals, np.ndarray) or normals.dtype not in (np.float32, np.float64):
        raise TypeError("normals must be a float32 or float64 ndarray")
    if vertices.ndim != 2 or vertices.shape[1] != 3 or faces.ndim != 2 or faces.shape[1] != 4:
        raise TypeError("vertices must be (N,3) and faces (M,4)")
    if normals.shape != (faces.shape[0], 3):
        raise TypeError("normals must be (M,3)")
    if not vertices.flags.c_contiguous or not vertices.flags.writeable:
        raise TypeError("vertices must be a writable C-contiguous array")  # PyReadwriteArray's requirement
    if int(n_iters) < 0:
        raise OverflowError("n_iters must fit a u32")
    if len(faces) and not np.all(faces[:, 0] == 3):
        raise ValueError("faces rows must start with the vertex count 3 (triangles)")
    f3 = np.ascontiguousarray(faces[:, 1:], dtype=np.int64)
    if len(f3) and (f3.min() < 0 or f3.max() >= len(vertices)):
        raise IndexError("face index out of bounds")  # the reference panics on the out-of-bounds row access
    f3 = f3.astype(np.int32)
    nrm = np.ascontiguousarray(normals, dtype=np.float64)
    L.check(L.lib().ivx_context_aware_smoothing(L.ptr(vertices), L.dtype_code(vertices, (L.F32, L.F64)),
                                                ctypes.c_int64(len(vertices)), L.ptr(f3), ctypes.c_int64(len(f3)),
                                                L.ptr(nrm), ctypes.c_double(t), ctypes.c_double(tmax),
                                                ctypes.c_double(bmin), ctypes.c_int(int(n_iters)), None, None),
            "context_aware_smoothing")


class Mesh:
    """invesalius_rs.Mesh (invesalius_rs/__init__.py:114-249) without the vtkPolyData constructor (no VTK on this
    path): built from arrays, or copied from another Mesh."""

    def __init__(self, pd=None, other=None, vertices=None, faces=None, normals=None):
        if pd is not None:
            raise TypeError("vtkPolyData input is outside this path; pass vertices, faces and normals")
        if other is not None:
            if not isinstance(other, Mesh):
                raise TypeError("other must be a Mesh instance")
            self._vertices = np.ascontiguousarray(other.vertices.copy())
            self._faces = np.ascontiguousarray(other.faces.copy())
            self._normals = np.ascontiguousarray(other.normals.copy())
        elif vertices is not None and faces is not None and normals is not None:
            self._vertices = np.ascontiguousarray(vertices)
            self._faces = np.ascontiguousarray(faces)
            self._normals = np.ascontiguousarray(normals)
        else:
            raise ValueError("Must provide either pd, other, or (vertices, faces, normals)")

    @classmethod
    def from_indexed(cls, verts, faces3):
        """(V,3) vertices + (T,3) faces of `surface_process.marching_cubes_indexed` -> Mesh with unit cell normals."""
        v = np.ascontiguousarray(verts)
        f3 = np.ascontiguousarray(faces3, dtype=np.int32)
        nrm = np.zeros((len(f3), 3), np.float64)
        if len(f3):
            L.check(L.lib().ivx_mesh_face_normals(L.ptr(v), L.dtype_code(v, (L.F32, L.F64)), ctypes.c_int64(len(v)),
                                                  L.ptr(f3), ctypes.c_int64(len(f3)), L.ptr(nrm)), "mesh_face_normals")
        f4 = np.empty((len(f3), 4), np.int64)
        f4[:, 0] = 3
        f4[:, 1:] = f3
        return cls(vertices=v, faces=f4, normals=nrm)

    @property
    def vertices(self):
        return self._vertices

    @property
    def faces(self):
        return self._faces

    @property
    def normals(self):
        return self._normals

    def ca_smoothing(self, T, tmax, bmin, n_iters):
        context_aware_smoothing(self._vertices, self._faces, self._normals, T, tmax, bmin, n_iters)


def ca_smoothing(mesh, T, tmax, bmin, n_iters):
    """invesalius_rs/__init__.py:251-275: smooths ``mesh`` in place."""
    mesh.ca_smoothing(T, tmax, bmin, n_iters)


def propagate_weights(vertices, faces3, seed_flags, tmax, bmin):
    """mesh.rs:204-288 on its own: weights from explicit seed flags (synchronous schedule, see k_smooth.hip)."""
    v = np.ascontiguousarray(vertices)
    f3 = np.ascontiguousarray(faces3, dtype=np.int32)
    s = np.ascontiguousarray(seed_flags, dtype=np.uint8)
    if s.shape != (len(v),):
        raise ValueError("one seed flag per vertex")
    w = np.zeros(len(v), np.float64)
    L.check(L.lib().ivx_mesh_propagate_weights(L.ptr(v), L.dtype_code(v, (L.F32, L.F64)), ctypes.c_int64(len(v)), L.ptr(f3),
                                               ctypes.c_int64(len(f3)), L.ptr(s), ctypes.c_double(tmax),
                                               ctypes.c_double(bmin), L.ptr(w)), "mesh_propagate_weights")
    return w


# ---------------------------------------------------------------------------------------------------------------------
# 3-D mask editing: mask_cut / brush_mask_rs / polygon2mask_rs (invesalius_rs/__init__.py:86-88) and count_regions
# (invesalius_rs/__init__.py:108-111)
# ---------------------------------------------------------------------------------------------------------------------
def _mat4(a, name):
    a = np.asarray(a)
    if a.dtype != np.float64 or a.ndim != 2:
        raise TypeError("%s must be a 2-D float64 array" % name)
    if a.shape != (4, 4):
        raise ValueError("%s must be 4x4" % name)  # Matrix4::from_row_slice panics on any other length
    return np.ascontiguousarray(a)


def mask_cut(image, sx, sy, sz, max_depth, mask, m, mv, out, edit_mode):
    """mask_cut_py.rs:9-69.  ``image`` is only type-checked (the reference never reads it); ``out`` uint8 (d,h,w) is
    edited in place; ``mask`` is the 2-D bool polygon filter, ``m`` / ``mv`` the world->screen and world->camera
    matrices."""
    if not isinstance(image, np.ndarray) or image.ndim != 3 or image.dtype not in (np.int16, np.uint8, np.float64):
        raise TypeError("Invalid image or mask type")
    if not isinstance(out, np.ndarray) or out.ndim != 3 or out.dtype != np.uint8:
        raise TypeError("Invalid image or mask type")
    if not isinstance(mask, np.ndarray) or mask.dtype != np.bool_ or mask.ndim != 2:
        raise TypeError("mask must be a 2-D bool array")
    if not out.flags.writeable:
        raise TypeError("out must be writable")
    m, mv = _mat4(m, "m"), _mat4(mv, "mv")
    mk = mask.view(np.uint8)
    L.check(L.lib().ivx_mask_cut(L.ptr(out), L.i64(out.shape), L.i64(out.strides), ctypes.c_double(sx), ctypes.c_double(sy),
                                 ctypes.c_double(sz), ctypes.c_double(max_depth), L.ptr(mk), ctypes.c_int64(mk.shape[0]),
                                 ctypes.c_int64(mk.shape[1]), L.i64(mk.strides), L.ptr(m), L.ptr(mv),
                                 ctypes.c_int(int(edit_mode))), "mask_cut")


def brush_mask_rs(out, orig, spacing, center, radius, edit_mode):
    """brush_mask_py.rs:8-28: spherical brush on a uint8 mask, in place.  edit_mode 1 erases, 0 reveals ``orig``
    (or paints 255 when ``orig`` is None)."""
    if not isinstance(out, np.ndarray) or out.ndim != 3 or out.dtype != np.uint8:
        raise TypeError("Invalid mask type for brush mask")
    if orig is not None and (not isinstance(orig, np.ndarray) or orig.ndim != 3 or orig.dtype != np.uint8):
        raise TypeError("orig must be a 3-D uint8 array or None")
    if orig is not None and orig.shape != out.shape:
        raise IndexError("orig and out differ in shape")  # orig_array[[z, y, x]] would panic
    if not out.flags.writeable:
        raise TypeError("out must be writable")
    sp = (ctypes.c_double * 3)(*[float(v) for v in spacing])
    ce = (ctypes.c_double * 3)(*[float(v) for v in center])
    L.check(L.lib().ivx_brush_mask(L.ptr(out), L.i64(out.shape), L.i64(out.strides), L.ptr(orig) if orig is not None else None,
                                   L.i64(orig.strides) if orig is not None else None, sp, ce, ctypes.c_double(radius),
                                   ctypes.c_int(int(edit_mode))), "brush_mask")


def polygon2mask_rs(shape, polygon):
    """polygon_mask_py.rs:7-27: ``shape`` = (w, h); ``polygon`` (N,2) float64 points; returns a (w,h) bool array."""
    w, h = (int(v) for v in shape)
    if w < 0 or h < 0:
        raise OverflowError("shape must be non-negative")
    if not isinstance(polygon, np.ndarray) or polygon.dtype != np.float64 or polygon.ndim != 2:
        raise TypeError("polygon must be a 2-D float64 array")
    if polygon.shape[0] and polygon.shape[1] < 2:
        raise IndexError("polygon rows need two coordinates")
    pts = np.ascontiguousarray(polygon[:, :2])
    out = np.zeros((w, h), np.uint8)
    L.check(L.lib().ivx_polygon2mask(ctypes.c_int64(w), ctypes.c_int64(h), L.ptr(pts), ctypes.c_int64(len(pts)), L.ptr(out)),
            "polygon2mask")
    return out.view(np.bool_)


def count_regions(image, number_regions):
    """invesalius_rs/__init__.py:108-111: uint32 array holding, for every voxel, the number of voxels that carry its
    label.  Labels are int16 / int32 / int64 in [0, number_regions]; anything else is where the reference panics."""
    if not isinstance(image, np.ndarray) or image.ndim != 3 or image.dtype not in (np.int16, np.int32, np.int64):
        raise TypeError("labels must be a 3-D int16, int32 or int64 array")
    if int(number_regions) < 0:
        raise OverflowError("number_regions must be non-negative")
    out = np.zeros(image.shape, np.uint32)
    L.check(L.lib().ivx_count_regions(L.dtype_code(image, (L.I16, L.I32, L.I64)), L.ptr(image), L.i64(image.shape),
                                      L.i64(image.strides), ctypes.c_int64(int(number_regions)), L.ptr(out)),
            "count_regions")  # IVX_ERANGE -> IndexError
    return out


def convolve_non_zero(volume, kernel, cval):
    """transforms_py.rs:51-93: float64 (d,h,w) volume correlated with a float64 kernel where the volume is non-zero
    (zero elsewhere); samples outside the volume count as ``cval`` (an i16 in the reference).  Returns a new array."""
    if not isinstance(volume, np.ndarray) or volume.dtype != np.float64 or volume.ndim != 3:
        raise TypeError("volume must be a 3-D float64 array")
    if not isinstance(kernel, np.ndarray) or kernel.dtype != np.float64 or kernel.ndim != 3:
        raise TypeError("kernel must be a 3-D float64 array")
    if isinstance(cval, float) and not float(cval).is_integer():
        raise TypeError("cval must be an integer (i16)")
    cval = int(cval)
    if not -32768 <= cval <= 32767:
        raise OverflowError("cval does not fit an i16")
    k = np.ascontiguousarray(kernel)
    out = np.zeros(volume.shape, np.float64)
    L.check(L.lib().ivx_convolve_non_zero(L.ptr(volume), L.i64(volume.shape), L.i64(volume.strides), L.ptr(k), L.i64(k.shape),
                                          ctypes.c_int(cval), L.ptr(out)), "convolve_non_zero")
    return out
