"""CPU ORACLE -- test infrastructure, NOT product code.

Python face of the oracle: ctypes bindings to ``oracle/libivx_oracle.so`` (the C restatement of the
reference's Rust kernels / third-party algorithms) plus numpy restatements of the pure-numpy reference
functions.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package ``invesalius3_amd`` never imports it and has no CPU fallback.

Reference citations are relative to the InVesalius checkout (``invesalius/...``, ``invesalius_rs/...``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import math

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DT = {np.dtype(np.uint8): 0, np.dtype(np.int16): 1, np.dtype(np.float64): 2, np.dtype(np.uint16): 3}


def build():
    """Compile the C oracle in place (gcc, a second or two)."""
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libivx_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_marching_cubes.restype = ctypes.c_int64
    return _LIB


def _i64(seq):
    return (ctypes.c_int64 * len(seq))(*[int(v) for v in seq])


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check(rc):
    if rc == -2:
        raise IndexError("oracle: seed/label out of bounds (reference: Rust index panic)")
    if rc == -4:
        raise ValueError("oracle: NumCast failure (reference: unwrap() panic)")
    if rc < 0:
        raise RuntimeError("oracle error %d" % rc)
    return rc


def _seeds(seeds):
    s = np.ascontiguousarray(np.array([tuple(x) for x in seeds], dtype=np.int64).reshape(-1, 3))
    return s


# ----------------------------------------------------------------------------------------------
# threshold                                  invesalius/data/slice_.py:1722-1769, 1225-1247
# ----------------------------------------------------------------------------------------------
def do_threshold_to_a_slice(slice_matrix, mask, threshold):
    """slice_.py:1722-1737 verbatim (intended dtype uint8 stated explicitly)."""
    thresh_min, thresh_max = threshold
    m = ((slice_matrix >= thresh_min) & (slice_matrix <= thresh_max)) * 255
    m[mask == 1] = 1
    m[mask == 2] = 2
    m[mask == 253] = 253
    m[mask == 254] = 254
    return m.astype("uint8")


def do_threshold_to_all_slices(mask_matrix, target_matrix, threshold_range):
    """slice_.py:1759-1767: slices whose flag mask[n,0,0] != 0 are skipped; flag set to 1 after."""
    for n in range(1, mask_matrix.shape[0]):
        if mask_matrix[n, 0, 0] == 0:
            m = mask_matrix[n, 1:, 1:]
            mask_matrix[n, 1:, 1:] = do_threshold_to_a_slice(target_matrix[n - 1], m, threshold_range)
            mask_matrix[n, 0, 0] = 1


def set_mask_threshold_volume(mask_matrix, image, threshold_range):
    """slice_.py:1240-1247 (whole-volume SetMaskThreshold: no preserve rule, flag forced to 1)."""
    thresh_min, thresh_max = threshold_range
    for n, slice_ in enumerate(image):
        m = np.ones(slice_.shape, mask_matrix.dtype)
        m[slice_ < thresh_min] = 0
        m[slice_ > thresh_max] = 0
        m[m == 1] = 255
        mask_matrix[n + 1, 1:, 1:] = m
        mask_matrix[n + 1, 0, 0] = 1


def set_mask_threshold_slice(slice_, threshold_range):
    """slice_.py:1253-1256 per-slice preview."""
    thresh_min, thresh_max = threshold_range
    return (255 * ((slice_ >= thresh_min) & (slice_ <= thresh_max))).astype("uint8")


# ----------------------------------------------------------------------------------------------
# window/level LUT                          invesalius/data/imagedata_utils.py:540-564
# np.piecewise allocates its result with the INPUT dtype, so int16 in -> int16 out with the float
# expression truncated toward zero.
# ----------------------------------------------------------------------------------------------
def get_LUT_value(data, window, level):
    shape = data.shape
    data_ = data.ravel()
    out = np.piecewise(
        data_,
        [data_ <= (level - 0.5 - (window - 1) / 2), data_ > (level - 0.5 + (window - 1) / 2)],
        [0, window, lambda d: ((d - (level - 0.5)) / (window - 1) + 0.5) * (window)],
    )
    out.shape = shape
    return out


def get_LUT_value_255(data, window, level):
    shape = data.shape
    data_ = data.ravel()
    out = np.piecewise(
        data_,
        [data_ <= (level - 0.5 - (window - 1) / 2), data_ > (level - 0.5 + (window - 1) / 2)],
        [0, 255, lambda d: ((d - (level - 0.5)) / (window - 1) + 0.5) * (255)],
    )
    out.shape = shape
    return out


# ----------------------------------------------------------------------------------------------
# floodfill family                                  invesalius_rs/src/floodfill.rs
# ----------------------------------------------------------------------------------------------
def floodfill_threshold(data, seeds, t0, t1, fill, strct, out):
    """generic_floodfill_threshold floodfill.rs:96-166 (wrapper semantics invesalius_rs/__init__.py:21-40)."""
    strct = np.ascontiguousarray(strct, dtype=np.uint8)
    if data.dtype.kind in "iu":
        t0, t1, fill = int(t0), int(t1), int(fill)
    s = _seeds(seeds)
    rc = lib().orc_floodfill_threshold(
        DT[data.dtype], _p(data), _i64(data.shape), _i64(data.strides), _p(s), ctypes.c_int64(len(s)),
        ctypes.c_double(t0), ctypes.c_double(t1), ctypes.c_int(int(fill)), _p(strct), _i64(strct.shape),
        _p(out), _i64(out.strides))
    _check(rc)


def floodfill_threshold_inplace(data, seeds, t0, t1, fill, strct):
    """generic_floodfill_threshold_inplace floodfill.rs:168-237."""
    strct = np.ascontiguousarray(strct, dtype=np.uint8)
    s = _seeds(seeds)
    rc = lib().orc_floodfill_threshold_inplace(
        DT[data.dtype], _p(data), _i64(data.shape), _i64(data.strides), _p(s), ctypes.c_int64(len(s)),
        ctypes.c_double(t0), ctypes.c_double(t1), ctypes.c_double(fill), _p(strct), _i64(strct.shape))
    _check(rc)


def floodfill(data, i, j, k, v, fill, out):
    """floodfill_internal floodfill.rs:5-49."""
    rc = lib().orc_floodfill(DT[data.dtype], _p(data), _i64(data.shape), _i64(data.strides),
                             ctypes.c_int64(i), ctypes.c_int64(j), ctypes.c_int64(k), ctypes.c_double(v),
                             ctypes.c_int(int(fill)), _p(out), _i64(out.strides))
    _check(rc)


def floodfill_auto_threshold(data, seeds, p, fill, out):
    """floodfill_auto_threshold floodfill_py.rs:12-85 (i16 only)."""
    assert data.dtype == np.int16
    s = _seeds(seeds)
    rc = lib().orc_floodfill_auto_threshold(_p(data), _i64(data.shape), _i64(data.strides), _p(s),
                                            ctypes.c_int64(len(s)), ctypes.c_float(p), ctypes.c_int(int(fill)),
                                            _p(out), _i64(out.strides))
    _check(rc)


def fill_holes_automatically(mask, labels, nlabels, max_size):
    """fill_holes_automatically_internal floodfill.rs:51-94."""
    assert mask.dtype == np.uint8 and labels.dtype == np.uint32
    rc = lib().orc_fill_holes(_p(mask), _i64(mask.shape), _i64(mask.strides), _p(labels), _i64(labels.strides),
                              ctypes.c_uint32(int(nlabels)), ctypes.c_uint32(int(max_size)))
    return bool(_check(rc))


def do_rg_confidence(image, p, bstruct, confid_mult, confid_iters):
    """invesalius/data/styles.py:3220-3251 (compute part; LUT branch left to the caller).
    Returns out_mask (uint8, fill=1).  Quirk Q4 kept: out_mask is NOT cleared between iterations;
    t0/t1 are floats and the wrapper truncates them with int() for integer images."""
    x, y, z = p
    bool_mask = np.zeros(image.shape, dtype="bool")
    out_mask = np.zeros(image.shape, dtype=np.uint8)
    for k in range(int(z - 1), int(z + 2)):
        if k < 0 or k >= bool_mask.shape[0]:
            continue
        for j in range(int(y - 1), int(y + 2)):
            if j < 0 or j >= bool_mask.shape[1]:
                continue
            for i in range(int(x - 1), int(x + 2)):
                if i < 0 or i >= bool_mask.shape[2]:
                    continue
                bool_mask[k, j, i] = True
    for _ in range(confid_iters):
        var = np.std(image[bool_mask])
        mean = np.mean(image[bool_mask])
        t0 = mean - var * confid_mult
        t1 = mean + var * confid_mult
        floodfill_threshold(image, ((x, y, z),), t0, t1, 1, bstruct, out_mask)
        bool_mask[out_mask == 1] = True
    return out_mask


# ----------------------------------------------------------------------------------------------
# projections                         invesalius/data/slice_.py:885-889 + invesalius_rs/src/mips.rs
# ----------------------------------------------------------------------------------------------
def maxip(a, axis):
    return np.array(a).max(axis)


def minip(a, axis):
    return np.array(a).min(axis)


def meanip(a, axis):
    return np.array(a).mean(axis)


def lmip(image, axis, tmin, tmax, out):
    """lmip mips.rs:7-86."""
    _check(lib().orc_lmip(DT[image.dtype], _p(image), _i64(image.shape), _i64(image.strides), int(axis),
                          ctypes.c_double(tmin), ctypes.c_double(tmax), _p(out), _i64(out.strides)))


def mida(image, axis, wl, ww, out):
    """mida_internal mips.rs:102-168; dtype pairs mips_py.rs:161-202."""
    _check(lib().orc_mida(DT[image.dtype], _p(image), _i64(image.shape), _i64(image.strides), int(axis),
                          ctypes.c_double(int(wl)), ctypes.c_double(int(ww)), DT[out.dtype], _p(out),
                          _i64(out.strides)))


def fcm_volume(image, n, axis):
    tmp = np.empty(image.shape, image.dtype)
    _check(lib().orc_fcm_volume(DT[image.dtype], _p(image), _i64(image.shape), _i64(image.strides),
                                ctypes.c_float(n), int(axis), _p(tmp)))
    return tmp


def powf_array(x, y):
    """libm's powf elementwise (float32 in, float32 out): Rust's f32::powf on this machine (mips.rs:211)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty(x.shape, np.float32)
    lib().orc_powf_array(_p(x), _p(y), _p(out), ctypes.c_int64(x.size))
    return out


def fast_countour_mip(image, n, axis, wl, ww, tmip, out):
    """fast_countour_mip_internal mips.rs:215-279."""
    _check(lib().orc_fast_countour_mip(DT[image.dtype], _p(image), _i64(image.shape), _i64(image.strides),
                                       ctypes.c_float(n), int(axis), ctypes.c_double(int(wl)),
                                       ctypes.c_double(int(ww)), int(tmip), _p(out), _i64(out.strides)))


# ----------------------------------------------------------------------------------------------
# marching cubes == create_surface_piece geometry    invesalius/data/surface_process.py:71-201
# ----------------------------------------------------------------------------------------------
def marching_cubes(a, spacing, iso_values, roi_start=0, pad_xy=True, pad_bottom=True, pad_top=True,
                   pad_value=0.0, vtk_pz=None):
    """Triangle soup float32 (T,3,3) of the padded + Y-flipped piece `a` (see ivx_oracle.c)."""
    if vtk_pz is None:
        vtk_pz = 1 if (pad_xy and pad_bottom) else 0
    iso = np.ascontiguousarray(iso_values, dtype=np.float64)
    sp = np.ascontiguousarray(spacing, dtype=np.float64)
    args = (DT[a.dtype], _p(a), _i64(a.shape), _i64(a.strides), int(bool(pad_xy)), int(bool(pad_bottom)),
            int(bool(pad_top)), ctypes.c_double(pad_value), int(vtk_pz), ctypes.c_int64(roi_start), _p(sp),
            _p(iso), len(iso))
    n = _check(lib().orc_marching_cubes(*args, None, ctypes.c_int64(0)))
    tris = np.empty((n, 3, 3), dtype=np.float32)
    if n:
        m = _check(lib().orc_marching_cubes(*args, _p(tris), ctypes.c_int64(n)))
        assert m == n
    return tris


def create_surface_piece(image, mask_matrix, roi, spacing, min_value, max_value, from_binary,
                         fill_border_holes=True):
    """Geometry of surface_process.py:100-186 for one piece: returns the triangle soup.
    `mask_matrix` is the (dz+1,dy+1,dx+1) mask; `roi` a slice over z of the IMAGE."""
    shape0 = image.shape[0] if image is not None else mask_matrix.shape[0] - 1
    pad_bottom = roi.start == 0
    pad_top = roi.stop >= shape0
    if from_binary:
        a = mask_matrix[roi.start + 1: roi.stop + 1, 1:, 1:]
        padv, isos = 0.0, [127.0]
    else:
        a = image[roi]
        padv, isos = float(np.iinfo(image.dtype).min), [float(min_value), float(max_value)]
    if fill_border_holes:
        return marching_cubes(a, spacing, isos, roi.start, True, pad_bottom, pad_top, padv, int(pad_bottom))
    return marching_cubes(a, spacing, isos, roi.start, False, False, False, padv, 0)


def write_stl_binary(path, tris):
    """vtkSTLWriter binary layout (surface.py:1827-1829): 80 B header, u32 count, 50 B/triangle."""
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    v = tris.astype(np.float64)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.where(ln > 0, ln, 1), 0.0)
    rec = np.zeros(len(tris), dtype=[("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    rec["n"] = n.astype(np.float32)
    rec["v"] = tris
    with open(path, "wb") as f:
        f.write(b"Visualization Toolkit generated SLA File".ljust(80))
        f.write(np.uint32(len(tris)).tobytes())
        f.write(rec.tobytes())


# ----------------------------------------------------------------------------------------------
# watershed                                  invesalius/data/watershed_process.py:19-60
# ----------------------------------------------------------------------------------------------
def watershed_ift(image, markers, strct):
    """scipy.ndimage.watershed_ift restated (ni_measure.c NI_WatershedIFT); pinned vs live scipy."""
    image = np.ascontiguousarray(image)
    markers = np.ascontiguousarray(markers)
    assert image.dtype in (np.uint8, np.uint16) and markers.dtype in (np.int8, np.int16)
    s3 = np.zeros((3, 3, 3), np.uint8)
    if image.ndim == 3:
        s3[:] = np.asarray(strct, dtype=np.uint8)
        shp = image.shape
    else:
        s3[1] = np.asarray(strct, dtype=np.uint8)
        shp = (1,) + image.shape
    out = np.empty_like(markers)
    _check(lib().orc_watershed_ift(0 if image.dtype == np.uint8 else 3, _p(image), _i64(shp),
                                   1 if markers.dtype == np.int16 else 4, _p(markers), _p(s3), _p(out)))
    return out


def watershed_merge(mask, tmp_mask, overwrite):
    """styles.py:2147-2152 (3-D) / 1984-1989 (2-D) merge rule, in place on `mask`."""
    if overwrite:
        mask[:] = 0
        mask[tmp_mask == 1] = 253
    else:
        sel = (mask == 0) | (mask == 2) | (mask == 253)
        mask[(tmp_mask == 2) & sel] = 2
        mask[(tmp_mask == 1) & sel] = 253


# ----------------------------------------------------------------------------------------------
# apply_view_matrix_transform               invesalius_rs/src/transforms_py.rs:12-147
# ----------------------------------------------------------------------------------------------
ORIENTATION = {"AXIAL": 0, "CORONAL": 1, "SAGITAL": 2}


def apply_view_matrix_transform(volume, spacing, m, n, orientation, minterpol, cval, out):
    """coord_transform over `out` (transforms.rs:9-55): nearest / trilinear / tricubic / Lanczos-4 resampling of
    `volume` through the 4x4 matrix `m` (row-major, C-contiguous float64)."""
    assert volume.dtype == out.dtype and volume.ndim == 3 and out.ndim == 3
    mm = np.ascontiguousarray(m, dtype=np.float64).reshape(16)
    sp = np.ascontiguousarray(spacing, dtype=np.float64)
    _check(lib().orc_apply_view_matrix_transform(
        DT[volume.dtype], _p(volume), _i64(volume.shape), _i64(volume.strides), _p(sp), _p(mm), ctypes.c_int64(int(n)),
        ORIENTATION.get(orientation, -1), int(minterpol), ctypes.c_double(float(cval)), _p(out), _i64(out.shape),
        _i64(out.strides)))


# ---------------------------------------------------------------------------------------------------------------------
# Surface post-processing (join_process_surface, invesalius/data/surface_process.py:376-391, 452-458).
# PARITY UNPINNED: the reference delegates both steps to VTK 9.3 (vtkPolyDataConnectivityFilter, vtkMassProperties),
# which is third party and not installed here.  What follows restates their published algorithms; the tests anchor
# them on size-independent properties instead (analytic volume/area of closed shapes, divergence theorem,
# scipy.sparse.csgraph components).
# ---------------------------------------------------------------------------------------------------------------------
def mesh_keep_largest(verts, faces):
    """Largest region by triangle count, first region (smallest first-triangle id) on a tie; triangles keep their
    order, vertices are compacted in order.  Returns (verts, faces, n_regions)."""
    v = np.asarray(verts, np.float32).reshape(-1, 3)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    if len(f) == 0:
        return v[:0].copy(), f[:0].astype(np.int32), 0
    parent = np.arange(len(v))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b, c in f:
        for q in (b, c):
            ra, rq = find(a), find(q)
            if ra != rq:
                parent[max(ra, rq)] = min(ra, rq)
    lab = np.array([find(a) for a in f[:, 0]])
    best, best_n = None, 0
    seen = {}
    for t, r in enumerate(lab):  # regions in order of their first triangle
        seen.setdefault(r, t)
    for r, _t in sorted(seen.items(), key=lambda kv: kv[1]):
        n = int((lab == r).sum())
        if n > best_n:
            best, best_n = r, n
    keep = lab == best
    kf = f[keep]
    used = np.zeros(len(v), bool)
    used[kf.ravel()] = True
    remap = np.cumsum(used) - 1
    return v[used].copy(), remap[kf].astype(np.int32), len(seen)


def mesh_mass_properties(verts, faces=None):
    """vtkMassProperties' algorithm (Alyassin et al. 1994) in float64, sequential sums:
    returns (volume, area, vol_x, vol_y, vol_z, kx, ky, kz)."""
    v = np.asarray(verts, np.float32).reshape(-1, 3).astype(np.float64)
    t = v.reshape(-1, 3, 3) if faces is None else v[np.asarray(faces, np.int64).reshape(-1, 3)]
    n = len(t)
    if n == 0:
        return (0.0,) * 8
    e0, e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0], t[:, 2] - t[:, 1]
    u = np.stack([e0[:, 1] * e1[:, 2] - e0[:, 2] * e1[:, 1], e0[:, 2] * e1[:, 0] - e0[:, 0] * e1[:, 2],
                  e0[:, 0] * e1[:, 1] - e0[:, 1] * e1[:, 0]], axis=1)
    ln = np.sqrt((u * u).sum(1))
    with np.errstate(invalid="ignore", divide="ignore"):
        u = np.where(ln[:, None] != 0.0, u / ln[:, None], 0.0)
    a0, a1, a2 = np.abs(u).T
    munc = [int(((a0 > a1) & (a0 > a2)).sum()), int(((a1 > a0) & (a1 > a2)).sum()), int(((a2 > a0) & (a2 > a1)).sum())]
    wxyz = int(((a0 == a1) & (a0 == a2)).sum())
    wxy = int(((a0 == a1) & (a0 > a2)).sum())
    wxz = int(((a0 == a2) & (a0 > a1)).sum())
    wyz = int(((a1 == a2) & (a0 < a2)).sum())
    a = np.sqrt((e1 * e1).sum(1))
    b = np.sqrt((e0 * e0).sum(1))
    c = np.sqrt((e2 * e2).sum(1))
    s = 0.5 * (a + b + c)
    area = np.sqrt(np.abs(s * (s - a) * (s - b) * (s - c)))
    avg = (t[:, 0] + t[:, 1] + t[:, 2]) / 3.0
    vol = [math.fsum(area * u[:, q] * avg[:, q]) for q in range(3)]
    kx = (munc[0] + wxyz / 3.0 + (wxy + wxz) / 2.0) / n
    ky = (munc[1] + wxyz / 3.0 + (wxy + wyz) / 2.0) / n
    kz = (munc[2] + wxyz / 3.0 + (wxz + wyz) / 2.0) / n
    return (abs(kx * vol[0] + ky * vol[1] + kz * vol[2]), math.fsum(area), vol[0], vol[1], vol[2], kx, ky, kz)


# ---------------------------------------------------------------------------------------------------------------------
# Context-aware smoothing: C restatement of invesalius_rs/src/mesh.rs (oracle/ivx_oracle_mesh.c)
# ---------------------------------------------------------------------------------------------------------------------
def context_aware_smoothing(vertices, faces4, normals, t, tmax, bmin, n_iters, details=False):
    """Smooths ``vertices`` (float32/float64, C-contiguous) IN PLACE like the reference; faces4 = (M,4) [3,v0,v1,v2].
    details=True also returns (staircase flags, weights)."""
    lib_ = lib()
    assert vertices.dtype in (np.float32, np.float64) and vertices.flags.c_contiguous
    f4 = np.ascontiguousarray(faces4, dtype=np.int64)
    nrm = np.ascontiguousarray(normals, dtype=np.float64)
    nv = len(vertices)
    flags = np.zeros(nv, np.uint8)
    w = np.zeros(nv, np.float64)
    lib_.orc_context_aware_smoothing.restype = ctypes.c_int
    rc = lib_.orc_context_aware_smoothing(ctypes.c_void_p(vertices.ctypes.data), ctypes.c_int(vertices.dtype == np.float64),
                                         ctypes.c_int64(nv), ctypes.c_void_p(f4.ctypes.data), ctypes.c_int64(len(f4)),
                                         ctypes.c_void_p(nrm.ctypes.data), ctypes.c_double(t), ctypes.c_double(tmax),
                                         ctypes.c_double(bmin), ctypes.c_int(n_iters), ctypes.c_void_p(flags.ctypes.data),
                                         ctypes.c_void_p(w.ctypes.data))
    if rc:
        raise RuntimeError("orc_context_aware_smoothing -> %d" % rc)
    return (flags, w) if details else None


def mesh_vertex_connectivity(faces3, nv):
    """build_vertex_connectivity (mesh.rs:102-121): CSR (off, idx) of unique neighbours in order of first appearance."""
    lib_ = lib()
    f3 = np.asarray(faces3, np.int64).reshape(-1, 3)
    f4 = np.empty((len(f3), 4), np.int64)
    f4[:, 0] = 3
    f4[:, 1:] = f3
    off = np.zeros(nv + 1, np.int64)
    lib_.orc_mesh_vertex_connectivity(ctypes.c_void_p(f4.ctypes.data), ctypes.c_int64(len(f4)), ctypes.c_int64(nv),
                                     ctypes.c_void_p(off.ctypes.data), None)
    idx = np.zeros(max(int(off[-1]), 1), np.int64)
    lib_.orc_mesh_vertex_connectivity(ctypes.c_void_p(f4.ctypes.data), ctypes.c_int64(len(f4)), ctypes.c_int64(nv),
                                     ctypes.c_void_p(off.ctypes.data), ctypes.c_void_p(idx.ctypes.data))
    return off, idx[: int(off[-1])]


def mesh_propagate_weights(vertices, faces3, seed_flags, tmax, bmin):
    lib_ = lib()
    v = np.ascontiguousarray(vertices)
    off, idx = mesh_vertex_connectivity(faces3, len(v))
    idx = np.ascontiguousarray(idx if len(idx) else np.zeros(1, np.int64))
    s = np.ascontiguousarray(seed_flags, dtype=np.uint8)
    w = np.zeros(len(v), np.float64)
    rc = lib_.orc_mesh_propagate_weights(ctypes.c_void_p(v.ctypes.data), ctypes.c_int(v.dtype == np.float64),
                                        ctypes.c_int64(len(v)), ctypes.c_void_p(off.ctypes.data),
                                        ctypes.c_void_p(idx.ctypes.data), ctypes.c_void_p(s.ctypes.data),
                                        ctypes.c_double(tmax), ctypes.c_double(bmin), ctypes.c_void_p(w.ctypes.data))
    if rc:
        raise RuntimeError("orc_mesh_propagate_weights -> %d" % rc)
    return w


def mesh_face_normals(vertices, faces3):
    v = np.asarray(vertices).astype(np.float64)
    f = np.asarray(faces3, np.int64).reshape(-1, 3)
    a, b = v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
    n = np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                  a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)
    ln = np.sqrt(n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2])
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(ln[:, None] != 0.0, n / ln[:, None], n)


# ---------------------------------------------------------------------------------------------------------------------
# 3-D mask editing kernels (oracle/ivx_oracle_edit.c)
# ---------------------------------------------------------------------------------------------------------------------
def mask_cut(out, sx, sy, sz, max_depth, mask, m, mv, edit_mode):
    """in place on a C-contiguous uint8 (d,h,w) array"""
    assert out.dtype == np.uint8 and out.flags.c_contiguous
    mk = np.ascontiguousarray(mask).view(np.uint8)
    m = np.ascontiguousarray(m, dtype=np.float64)
    mv = np.ascontiguousarray(mv, dtype=np.float64)
    lib().orc_mask_cut(ctypes.c_double(sx), ctypes.c_double(sy), ctypes.c_double(sz), ctypes.c_double(max_depth),
                       ctypes.c_void_p(mk.ctypes.data), ctypes.c_int64(mk.shape[0]), ctypes.c_int64(mk.shape[1]),
                       ctypes.c_void_p(m.ctypes.data), ctypes.c_void_p(mv.ctypes.data), ctypes.c_void_p(out.ctypes.data),
                       _i64(out.shape), ctypes.c_int(edit_mode))


def brush_mask(out, orig, spacing, center, radius, edit_mode):
    assert out.dtype == np.uint8 and out.flags.c_contiguous
    o = np.ascontiguousarray(orig) if orig is not None else None
    lib().orc_brush_mask(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(o.ctypes.data) if o is not None else None,
                         _i64(out.shape), (ctypes.c_double * 3)(*spacing), (ctypes.c_double * 3)(*center),
                         ctypes.c_double(radius), ctypes.c_int(edit_mode))


def polygon2mask(shape, polygon):
    w, h = shape
    pts = np.ascontiguousarray(polygon, dtype=np.float64).reshape(-1, 2)
    out = np.zeros((w, h), np.uint8)
    lib().orc_polygon2mask(ctypes.c_int64(w), ctypes.c_int64(h), ctypes.c_void_p(pts.ctypes.data), ctypes.c_int64(len(pts)),
                           ctypes.c_void_p(out.ctypes.data))
    return out.view(np.bool_)


def count_regions(image, number_regions):
    lab = np.ascontiguousarray(image, dtype=np.int64)
    out = np.zeros(lab.shape, np.uint32)
    rc = lib().orc_count_regions(ctypes.c_void_p(lab.ctypes.data), ctypes.c_int64(lab.size), ctypes.c_int64(number_regions),
                                 ctypes.c_void_p(out.ctypes.data))
    if rc == -1:
        raise IndexError("label out of range")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# convolve_non_zero (invesalius_rs/src/transforms_py.rs:51-93) and Slice.calc_image_area (slice_.py:2296-2322)
# ---------------------------------------------------------------------------------------------------------------------
def convolve_non_zero(volume, kernel, cval):
    """term-by-term restatement in numpy: the (k, j, i) accumulation order is kept by adding one shifted plane at a time"""
    v = np.asarray(volume, np.float64)
    k = np.asarray(kernel, np.float64)
    pz, py, px = (s // 2 for s in k.shape)
    pad = np.pad(v, ((pz, k.shape[0] - 1 - pz), (py, k.shape[1] - 1 - py), (px, k.shape[2] - 1 - px)), constant_values=float(int(cval)))
    acc = np.zeros(v.shape, np.float64)
    for a in range(k.shape[0]):
        for b in range(k.shape[1]):
            for c in range(k.shape[2]):
                acc = acc + pad[a:a + v.shape[0], b:b + v.shape[1], c:c + v.shape[2]] * k[a, b, c]
    return np.where(v != 0.0, acc, 0.0)


def calc_image_area(mask_matrix, spacing):
    sx, sy, sz = (float(s) for s in spacing)
    kernel = np.zeros((3, 3, 3))
    kernel[1, 1, 1] = 2 * sx * sy + 2 * sx * sz + 2 * sy * sz
    kernel[0, 1, 1] = -(sx * sy)
    kernel[2, 1, 1] = -(sx * sy)
    kernel[1, 0, 1] = -(sx * sz)
    kernel[1, 2, 1] = -(sx * sz)
    kernel[1, 1, 0] = -(sy * sz)
    kernel[1, 1, 2] = -(sy * sz)
    bin_img = mask_matrix[1:, 1:, 1:] > 127
    return float(convolve_non_zero(bin_img * 1.0, kernel, 1).sum())


def mask_fill_holes_auto(matrix, target, conn, orientation, index, size):
    """Mask.fill_holes_auto (invesalius/data/mask.py:519-562) minus the history: scipy.ndimage.label (the third-party
    function the reference calls) + the fill_holes_automatically restatement.  In place; returns the bool."""
    from scipy import ndimage
    if target == "3D":
        view = matrix[1:, 1:, 1:]
        bstruct = ndimage.generate_binary_structure(3, {6: 1, 18: 2, 26: 3}[conn])
    else:
        view = {"AXIAL": lambda: matrix[index + 1, 1:, 1:], "CORONAL": lambda: matrix[1:, index + 1, 1:],
                "SAGITAL": lambda: matrix[1:, 1:, index + 1]}[orientation]()
        bstruct = ndimage.generate_binary_structure(2, {4: 1, 8: 2}[conn])
    imask = ~(view > 127)
    labels, nlabels = ndimage.label(imask, bstruct, output=np.uint32)
    labels = np.asarray(labels, dtype=np.uint32, order="C")
    if nlabels == 0:
        return False
    if target != "3D":
        labels = labels.reshape(1, labels.shape[0], labels.shape[1])
        work = np.ascontiguousarray(view).reshape(1, view.shape[0], view.shape[1])
        ret = fill_holes_automatically(work, labels, nlabels, size)
        view[...] = work[0]
        return ret
    work = np.ascontiguousarray(view)
    ret = fill_holes_automatically(work, labels, nlabels, size)
    view[...] = work
    return ret


def jump_flooding(distance_map, map_owners, sites, normalize):
    """jump_flooding_internal floodfill.rs:298-507 (wrapper invesalius_rs/__init__.py:76-80): float32 distances and int32
    owners (1-based site index, 0 = none) updated in place; sites = (n, 3) int32 rows of (z, y, x)."""
    assert distance_map.dtype == np.float32 and map_owners.dtype == np.int32
    assert distance_map.flags.c_contiguous and map_owners.flags.c_contiguous and distance_map.shape == map_owners.shape
    s = np.ascontiguousarray(sites, dtype=np.int32).reshape(-1, 3)
    rc = lib().orc_jump_flooding(_p(distance_map), _p(map_owners), _i64(distance_map.shape), _p(s),
                                 ctypes.c_int64(len(s)), ctypes.c_int(1 if normalize else 0))
    _check(rc)


def _ws_args(image, markers, strct):
    image = np.ascontiguousarray(image)
    markers = np.ascontiguousarray(markers)
    assert image.dtype in (np.uint8, np.uint16) and markers.dtype in (np.int8, np.int16)
    s3 = np.zeros((3, 3, 3), np.uint8)
    if image.ndim == 3:
        s3[:] = np.asarray(strct, dtype=np.uint8)
        shp = image.shape
    else:
        s3[1] = np.asarray(strct, dtype=np.uint8)
        shp = (1,) + image.shape
    return image, markers, s3, shp


def watershed_ift_events(image, markers, strct):
    """orc_watershed_ift (== live scipy) plus the event counts of scipy's linked-list defect:
    (sole-element re-queues, late pops, double pops, never popped)."""
    image, markers, s3, shp = _ws_args(image, markers, strct)
    out = np.empty_like(markers)
    ev = (ctypes.c_int64 * 4)()
    _check(lib().orc_watershed_ift_events(0 if image.dtype == np.uint8 else 3, _p(image), _i64(shp),
                                          1 if markers.dtype == np.int16 else 4, _p(markers), _p(s3), _p(out), ev))
    return out, tuple(int(v) for v in ev)


def watershed_ift_trace(image, markers, strct):
    """watershed_ift_events plus WHERE: per-voxel flags (1 popped late, 2 never popped, 4 trigger, 8 popped twice) and
    per-level counters (pops, late pops, triggers), each of length 65536.  The event tuple has two more entries: unlinks
    that spliced through a predecessor / a successor which did not link to the voxel (the only way the defect does harm)."""
    image, markers, s3, shp = _ws_args(image, markers, strct)
    out = np.empty_like(markers)
    ev = (ctypes.c_int64 * 6)()
    flags = np.zeros(image.shape, np.uint8)
    lvl = np.zeros((3, 65536), np.int64)
    _check(lib().orc_watershed_ift_trace(0 if image.dtype == np.uint8 else 3, _p(image), _i64(shp),
                                         1 if markers.dtype == np.int16 else 4, _p(markers), _p(s3), _p(out), ev, _p(flags), _p(lvl)))
    return out, tuple(int(v) for v in ev), flags, lvl


def watershed_ift_clean(image, markers, strct, want_cost=False):
    """The algorithm watershed_ift documents, without the linked-list defect (ivx_oracle_wsz.c)."""
    image, markers, s3, shp = _ws_args(image, markers, strct)
    out = np.empty_like(markers)
    cost = np.empty(image.shape, np.uint32) if want_cost else None
    _check(lib().orc_watershed_ift_clean(0 if image.dtype == np.uint8 else 3, _p(image), _i64(shp),
                                         1 if markers.dtype == np.int16 else 4, _p(markers), _p(s3), _p(out),
                                         _p(cost) if want_cost else None))
    return (out, cost) if want_cost else out


def watershed_sk(image, markers, strct, tie_mode=0, want_stats=False):
    """skimage.segmentation.watershed(image, markers, strct) as the reference calls it (watershed_process.py:39,52),
    restated in ivx_oracle_wssk.c.  tie_mode 0 = scikit-image's binary heap move for move (pinned to the compiled
    0.18.3 kernel, tests/golden/watershed_sk.npz); 1 = equal-valued marker voxels leave the queue in raster order.
    Returns int32 labels (scikit-image's output dtype)."""
    image = np.ascontiguousarray(image)
    markers = np.ascontiguousarray(markers)
    idt = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 3, np.dtype(np.int16): 1}[image.dtype]
    mdt = {np.dtype(np.int16): 1, np.dtype(np.int8): 4, np.dtype(np.int32): 2}[markers.dtype]
    s3 = np.zeros((3, 3, 3), np.uint8)
    if image.ndim == 3:
        s3[:] = np.asarray(strct, dtype=np.uint8)
        shp = image.shape
    else:
        s3[1] = np.asarray(strct, dtype=np.uint8)
        shp = (1,) + image.shape
    out = np.zeros(image.shape, np.int32)
    st = np.zeros(4, np.int64)
    _check(lib().orc_watershed_sk(idt, _p(image), _i64(shp), mdt, _p(markers), _p(s3), ctypes.c_int(int(tie_mode)),
                                  _p(out), _p(st)))
    if want_stats:
        return out, {"pushes": int(st[0]), "pops": int(st[1]), "heap_peak": int(st[2]), "tied_marker_pops": int(st[3])}
    return out
