"""Geometry of ``invesalius.data.surface_process.create_surface_piece`` on the GPU.

The reference pads the piece (``pad_image`` surface_process.py:52-68), wraps it as vtkImageData with the extent
conventions of ``converters.to_vtk`` (converters.py:34-101), flips Y about the origin and contours it with
``vtkContourFilter`` (surface_process.py:156-186), then writes a ``.vtp``.  Here padding and flip are folded into
the kernel's addressing and the result is the triangle soup itself (float32 ``(T, 3, 3)``); ``write_stl_binary``
is the vtkSTLWriter-compatible sink (surface.py:1827-1829).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib as L


def marching_cubes(a, spacing, iso_values, roi_start=0, pad_xy=True, pad_bottom=True, pad_top=True, pad_value=0.0,
                   vtk_pz=None) -> np.ndarray:
    """Triangle soup of the (virtually) padded + Y-flipped piece ``a`` (3-D uint8 / int16 / uint16)."""
    if a.ndim != 3:
        raise TypeError("piece must be 3-D")
    code = L.dtype_code(a, (L.U8, L.I16, L.U16))
    iso = [float(v) for v in iso_values]
    if not 1 <= len(iso) <= 2:
        raise ValueError("one or two iso-values")
    if vtk_pz is None:
        vtk_pz = 1 if (pad_xy and pad_bottom) else 0
    p = L.McParams()
    p.dtype, p.pad_xy, p.pad_bottom, p.pad_top = code, int(bool(pad_xy)), int(bool(pad_bottom)), int(bool(pad_top))
    p.vtk_pz, p.niso = int(vtk_pz), len(iso)
    p.nz, p.ny, p.nx = a.shape
    p.roi_start = int(roi_start)
    p.pad_value = float(pad_value)
    p.spacing[:] = [float(s) for s in spacing]
    p.iso[:] = iso + [0.0] * (2 - len(iso))
    n = ctypes.c_int64(0)
    lib = L.lib()
    L.check(lib.ivx_marching_cubes(ctypes.byref(p), L.ptr(a), L.i64(a.strides), None, ctypes.c_int64(0),
                                   ctypes.byref(n)), "marching_cubes")
    tris = np.empty((n.value, 3, 3), np.float32)
    if n.value:
        m = ctypes.c_int64(0)
        L.check(lib.ivx_marching_cubes(ctypes.byref(p), L.ptr(a), L.i64(a.strides), L.ptr(tris),
                                       ctypes.c_int64(n.value), ctypes.byref(m)), "marching_cubes")
        assert m.value == n.value
    return tris


def _mc_params(a, spacing, iso_values, roi_start, pad_xy, pad_bottom, pad_top, pad_value, vtk_pz):
    code = L.dtype_code(a, (L.U8, L.I16, L.U16))
    iso = [float(v) for v in iso_values]
    if not 1 <= len(iso) <= 2:
        raise ValueError("one or two iso-values")
    if vtk_pz is None:
        vtk_pz = 1 if (pad_xy and pad_bottom) else 0
    p = L.McParams()
    p.dtype, p.pad_xy, p.pad_bottom, p.pad_top = code, int(bool(pad_xy)), int(bool(pad_bottom)), int(bool(pad_top))
    p.vtk_pz, p.niso = int(vtk_pz), len(iso)
    p.nz, p.ny, p.nx = a.shape
    p.roi_start = int(roi_start)
    p.pad_value = float(pad_value)
    p.spacing[:] = [float(s) for s in spacing]
    p.iso[:] = iso + [0.0] * (2 - len(iso))
    return p


def marching_cubes_indexed(a, spacing, iso_values, roi_start=0, pad_xy=True, pad_bottom=True, pad_top=True, pad_value=0.0,
                           vtk_pz=None):
    """The same surface as `marching_cubes`, with coincident points merged: returns ``(verts (V,3) float32,
    faces (T,3) int32)`` such that ``verts[faces]`` is the soup, triangle for triangle.  This is what
    join_process_surface gets out of vtkAppendPolyData + vtkCleanPolyData (surface_process.py:229-268)."""
    if a.ndim != 3:
        raise TypeError("piece must be 3-D")
    p = _mc_params(a, spacing, iso_values, roi_start, pad_xy, pad_bottom, pad_top, pad_value, vtk_pz)
    nv, nt = ctypes.c_int64(0), ctypes.c_int64(0)
    lib = L.lib()
    L.check(lib.ivx_marching_cubes_indexed(ctypes.byref(p), L.ptr(a), L.i64(a.strides), None, ctypes.c_int64(0), None,
                                           ctypes.c_int64(0), ctypes.byref(nv), ctypes.byref(nt)), "marching_cubes_indexed")
    verts = np.empty((nv.value, 3), np.float32)
    faces = np.empty((nt.value, 3), np.int32)
    if nt.value:
        L.check(lib.ivx_marching_cubes_indexed(ctypes.byref(p), L.ptr(a), L.i64(a.strides), L.ptr(verts),
                                               ctypes.c_int64(nv.value), L.ptr(faces), ctypes.c_int64(nt.value),
                                               ctypes.byref(nv), ctypes.byref(nt)), "marching_cubes_indexed")
    return verts, faces


def mass_properties(verts, faces=None):
    """``(volume, area)`` of a triangle mesh, the two numbers join_process_surface reports through
    vtkMassProperties (surface_process.py:452-458).  ``faces=None`` takes ``verts`` as a (T,3,3) soup."""
    return mass_properties_full(verts, faces)[:2]


def mass_properties_full(verts, faces=None):
    """(volume, area, vol_x, vol_y, vol_z, kx, ky, kz): the weighted discrete-divergence terms behind the volume."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(8, np.float64)
    if faces is None:
        if len(v) % 3:
            raise ValueError("a soup has 3 vertices per triangle")
        f, nt = None, len(v) // 3
    else:
        f = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        nt = len(f)
    L.check(L.lib().ivx_mesh_mass_properties(L.ptr(v), ctypes.c_int64(len(v)), L.ptr(f) if f is not None else None,
                                             ctypes.c_int64(nt), L.ptr(out)), "mesh_mass_properties")
    return tuple(float(x) for x in out)


def keep_largest(verts, faces):
    """Largest connected region of an indexed mesh (vtkPolyDataConnectivityFilter, LargestRegion;
    surface_process.py:376-391): most triangles, earliest first triangle on a tie.  Returns
    ``(verts', faces', n_regions)`` with the kept triangles in their original order and the vertices compacted."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
    ov = np.empty_like(v)
    of = np.empty_like(f)
    nv, nt, nr = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    L.check(L.lib().ivx_mesh_keep_largest(L.ptr(v), ctypes.c_int64(len(v)), L.ptr(f), ctypes.c_int64(len(f)), L.ptr(ov),
                                          L.ptr(of), ctypes.byref(nv), ctypes.byref(nt), ctypes.byref(nr)),
            "mesh_keep_largest")
    return ov[: nv.value].copy(), of[: nt.value].copy(), nr.value


def create_surface_piece(image, mask_matrix, roi, spacing, min_value, max_value, from_binary,
                         fill_border_holes=True) -> np.ndarray:
    """One piece of surface_process.py:71-201 (arguments reduced to the ones that reach the geometry).

    ``mask_matrix`` is the ``(dz+1,dy+1,dx+1)`` mask; ``roi`` a ``slice`` over the IMAGE z axis
    (surface.py:1378-1380).  from_binary contours ``mask[roi+1, 1:, 1:]`` at 127; otherwise the raw image at
    ``min_value`` and ``max_value``."""
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


def create_surface(image, mask_matrix, spacing, min_value, max_value, from_binary, fill_border_holes=True,
                   piece_size=20, o_piece=1):
    """SurfaceManager.AddNewActor's piece loop (surface.py:1362-1380): 20-slice pieces + 1 overlap slice,
    concatenated (vtkAppendPolyData without the point merge)."""
    dz = image.shape[0] if image is not None else mask_matrix.shape[0] - 1
    n_pieces = int(round(dz / piece_size + 0.5, 0))
    parts = []
    for i in range(n_pieces):
        roi = slice(i * piece_size, i * piece_size + piece_size + o_piece)
        if roi.start >= dz:
            break
        parts.append(create_surface_piece(image, mask_matrix, roi, spacing, min_value, max_value, from_binary,
                                          fill_border_holes))
    return np.concatenate(parts) if parts else np.empty((0, 3, 3), np.float32)


def join_process_surface(image, mask_matrix, spacing, min_value, max_value, from_binary, fill_border_holes=True,
                         keep_largest_region=False):
    """The geometry of join_process_surface (surface_process.py:204-472) for the stages built here: the pieces'
    surfaces appended and point-merged (== one indexed marching-cubes pass over the whole volume: pieces share
    exactly one slice, so their cells partition the volume's), optionally the largest region only, then area and
    volume.  Returns ``(verts, faces, {"volume": v, "area": a})``.  Smoothing, decimation, hole filling and normals
    (VTK filters in the reference) are not part of this path."""
    dz = image.shape[0] if image is not None else mask_matrix.shape[0] - 1
    if from_binary:
        a = mask_matrix[1: dz + 1, 1:, 1:]
        padv, isos = 0.0, [127.0]
    else:
        a = image
        padv, isos = float(np.iinfo(image.dtype).min), [float(min_value), float(max_value)]
    if fill_border_holes:
        verts, faces = marching_cubes_indexed(a, spacing, isos, 0, True, True, True, padv, 1)
    else:
        verts, faces = marching_cubes_indexed(a, spacing, isos, 0, False, False, False, padv, 0)
    if keep_largest_region and len(faces):
        verts, faces, _ = keep_largest(verts, faces)
    volume, area = mass_properties(verts, faces)
    return verts, faces, {"volume": volume, "area": area}


def write_stl_binary(path, tris):
    """vtkSTLWriter binary layout (surface.py:1827-1829): 80-byte header, u32 count, 50 bytes per triangle."""
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    v = tris.astype(np.float64)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.divide(n, ln, out=np.zeros_like(n), where=ln > 0)
    rec = np.zeros(len(tris), dtype=[("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    rec["n"] = n.astype(np.float32)
    rec["v"] = tris
    with open(path, "wb") as f:
        f.write(b"Visualization Toolkit generated SLA File".ljust(80))
        f.write(np.uint32(len(tris)).tobytes())
        f.write(rec.tobytes())
