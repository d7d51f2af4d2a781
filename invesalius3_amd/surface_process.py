"""Geometry of ``invesalius.data.surface_process.create_surface_piece`` on the GPU.

The reference pads the piece (``pad_image`` surface_process.py:52-68), wraps it as vtkImageData with the extent
conventions of ``converters.to_vtk`` (converters.py:34-101), flips Y about the origin and contours it with
``vtkContourFilter`` (surface_process.py:156-186), then writes a ``.vtp``.  Here padding and flip are folded into
the kernel's addressing.  ``create_surface_piece`` keeps the reference's 20-argument signature (memmap file names in,
``.vtp`` file name out -- what ``SurfaceManager.AddNewActor`` passes, surface.py:1381-1410); ``surface_piece`` is the
same geometry on arrays and returns the float32 ``(T, 3, 3)`` soup; ``write_stl_binary`` is the vtkSTLWriter-compatible
sink (surface.py:1827-1829).
"""
from __future__ import annotations

import base64
import ctypes
import os
import re
import tempfile

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
    # one upload, one count, one emit: the soup waits in the library's output block while numpy makes room for it (round 6; the
    # count call + emit call below sent a 512^3 mask over PCIe twice -- 14 ms of which 0.15 ms were kernels)
    L.check(lib.ivx_marching_cubes_begin(ctypes.byref(p), L.ptr(a), L.i64(a.strides), ctypes.byref(n)), "marching_cubes")
    tris = np.empty((n.value, 3, 3), np.float32)
    rc = lib.ivx_marching_cubes_fetch(L.ptr(tris), ctypes.c_int64(n.value))
    if rc == L.IVX_OK:
        return tris
    if rc != L.IVX_EINVAL:
        L.check(rc, "marching_cubes")
    # (another thread's host-level call came in between the two halves: the two-call form)
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


def surface_piece(image, mask_matrix, roi, spacing, min_value, max_value, from_binary,
                  fill_border_holes=True) -> np.ndarray:
    """The geometry of one piece of surface_process.py:71-201 on arrays already in memory (the arguments that reach it).

    ``mask_matrix`` is the ``(dz+1,dy+1,dx+1)`` mask; ``roi`` a ``slice`` over the IMAGE z axis
    (surface.py:1378-1380).  from_binary contours ``mask[roi+1, 1:, 1:]`` at 127; otherwise the raw image at
    ``min_value`` and ``max_value``."""
    shape0 = image.shape[0] if image is not None else mask_matrix.shape[0] - 1
    return _surface_piece(image, mask_matrix, roi, spacing, min_value, max_value, from_binary, fill_border_holes, shape0)


def _surface_piece(image, mask_matrix, roi, spacing, min_value, max_value, from_binary, fill_border_holes, shape0):
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


# ---- VTK XML PolyData (.vtp), the piece format of the reference (vtkXMLPolyDataWriter, surface_process.py:193-196) ----
def write_vtp(path, verts, faces, point_normals=None, cell_normals=None):
    """Indexed triangles as an uncompressed inline-binary ``.vtp``: per DataArray, base64 of a UInt32 byte count followed by
    the raw little-endian data -- the layout vtkXMLPolyDataReader (surface_process.py:237-243) parses.  VTK is not
    installed in this environment, so the file is checked against the format's documentation and our own reader only.
    `point_normals` / `cell_normals` go in as the active "Normals" arrays of PointData / CellData (what vtkPolyDataNormals
    leaves in the file join_process_surface writes, :420-435)."""
    v = np.ascontiguousarray(verts, dtype="<f4").reshape(-1, 3)
    f = np.ascontiguousarray(faces, dtype="<i4").reshape(-1, 3)

    def arr(a, name, ncomp=None):
        raw = a.tobytes()
        text = base64.b64encode(np.uint32(len(raw)).tobytes() + raw).decode()
        t = {"<f4": "Float32", "<i4": "Int32"}[a.dtype.str]
        comp = ' NumberOfComponents="%d"' % ncomp if ncomp else ""
        return '<DataArray type="%s" Name="%s"%s format="binary">%s</DataArray>' % (t, name, comp, text)

    offsets = (np.arange(1, len(f) + 1, dtype="<i4") * 3)
    extra = ""
    if point_normals is not None:
        extra += '<PointData Normals="Normals">%s</PointData>\n' % arr(np.ascontiguousarray(point_normals, dtype="<f4").reshape(-1, 3), "Normals", 3)
    if cell_normals is not None:
        extra += '<CellData Normals="Normals">%s</CellData>\n' % arr(np.ascontiguousarray(cell_normals, dtype="<f4").reshape(-1, 3), "Normals", 3)
    xml = ('<?xml version="1.0"?>\n<VTKFile type="PolyData" version="0.1" byte_order="LittleEndian" header_type="UInt32">\n'
           '<PolyData>\n<Piece NumberOfPoints="%d" NumberOfVerts="0" NumberOfLines="0" NumberOfStrips="0" NumberOfPolys="%d">\n'
           '%s<Points>%s</Points>\n<Polys>%s%s</Polys>\n</Piece>\n</PolyData>\n</VTKFile>\n'
           % (len(v), len(f), extra, arr(v, "Points", 3), arr(f.reshape(-1), "connectivity"), arr(offsets, "offsets")))
    with open(path, "w") as fh:
        fh.write(xml)


def read_vtp(path):
    """Reader for the files `write_vtp` makes -> ``(verts (V,3) float32, faces (T,3) int32)`` (triangles only)."""
    txt = open(path).read()
    out = {}
    for m in re.finditer(r'<DataArray type="(\w+)" Name="(\w+)"[^>]*format="binary">([^<]*)</DataArray>', txt):
        raw = base64.b64decode(m.group(3))
        n = int(np.frombuffer(raw[:4], "<u4")[0])
        out[m.group(2)] = np.frombuffer(raw[4:4 + n], {"Float32": "<f4", "Int32": "<i4"}[m.group(1)]).copy()
    verts = out.get("Points", np.zeros(0, np.float32)).reshape(-1, 3)
    conn = out.get("connectivity", np.zeros(0, np.int32))
    off = out.get("offsets", np.zeros(0, np.int32))
    if len(off) and not np.array_equal(off, np.arange(1, len(off) + 1, dtype=np.int32) * 3):
        raise ValueError("%s: only triangle pieces are supported" % path)
    return verts, conn.reshape(-1, 3)


def create_surface_piece(filename, shape, dtype, mask_filename, mask_shape, mask_dtype, roi, spacing, mode, min_value,
                         max_value, decimate_reduction, smooth_relaxation_factor, smooth_iterations, language, flip_image,
                         from_binary, algorithm, imagedata_resolution, fill_border_holes):
    """``invesalius.data.surface_process.create_surface_piece`` (:71-201) with its own signature: opens the image and
    mask memmaps by file name, contours the piece ``roi`` on the GPU and writes the surface to a ``.vtp`` whose name it
    returns (``tempfile.mkstemp(suffix="_%d_%d.vtp" % (roi.start, roi.stop))``, :192).

    Like the reference's body, `mode`, the decimate / smooth numbers, `language`, `flip_image` and
    `imagedata_resolution` are accepted and unused here (the flip is unconditional, :156-161; the resampling of the
    Low / Medium presets happens in the caller, surface.py:1350-1353 -> `resize_image_array` below).  ``from_binary``
    contours ``mask[roi + 1, 1:, 1:]`` at 127; otherwise the raw image at `min_value` and `max_value` -- after the
    ``"InVesalius 3.b2"`` value rewrite (:128-146) when that algorithm is named: voxels the mask marks 1 drop below the
    image minimum, voxels marked 254 become the mid threshold; the vtkImageGaussianSmooth that follows there has radius
    factor 0.3 at VTK's default standard deviation 2 -> kernel radius int(0.6) = 0, i.e. the identity (parity unpinned:
    VTK is not installed; the branch is unreachable from AddNewActor, which passes from_binary=True for it)."""
    roi = roi if isinstance(roi, slice) else slice(*roi)
    mask = np.memmap(mask_filename, mode="r", dtype=mask_dtype, shape=tuple(mask_shape))
    shape = tuple(shape)
    pad_bottom = roi.start == 0
    pad_top = roi.stop >= shape[0]
    if from_binary:
        image = None
    else:
        image = np.memmap(filename, mode="r", dtype=dtype, shape=shape)
        if image.dtype != np.int16:
            raise TypeError("the image memmap must be int16")
    if not from_binary and algorithm == "InVesalius 3.b2":
        a_image = np.array(image[roi])
        a_mask = np.array(mask[roi.start + 1: roi.stop + 1, 1:, 1:])
        # (with fill_border_holes the reference indexes its PADDED piece with the unpadded mask and raises IndexError: dead
        #  code there; here the rewrite is applied to the piece itself and the padding stays virtual)
        a_image[a_mask == 1] = np.array(int(a_image.min()) - 1).astype(a_image.dtype)
        a_image[a_mask == 254] = (min_value + max_value) / 2.0
        padv = float(np.iinfo(a_image.dtype).min)
        tris = (marching_cubes(a_image, spacing, [float(min_value), float(max_value)], roi.start, True, pad_bottom, pad_top, padv,
                               int(pad_bottom)) if fill_border_holes else
                marching_cubes(a_image, spacing, [float(min_value), float(max_value)], roi.start, False, False, False, padv, 0))
    else:
        dz = shape[0]
        tris = _surface_piece(image, mask, roi, spacing, min_value, max_value, from_binary, fill_border_holes, dz)
    verts = tris.reshape(-1, 3)
    faces = np.arange(len(verts), dtype=np.int32).reshape(-1, 3)
    fd, out_name = tempfile.mkstemp(suffix="_%d_%d.vtp" % (roi.start, roi.stop))
    os.close(fd)
    write_vtp(out_name, verts, faces)
    return out_name


def join_surface_pieces(filenames, keep_largest_region=False):
    """The stages of join_process_surface (surface_process.py:204-472) that are built here, on piece FILES: read the
    ``.vtp`` pieces in z order (:229-250 sorts by the roi in the name), append them, merge coincident points
    (vtkCleanPolyData: exact float32 coincidence, as between two pieces' copies of their shared plane), optionally keep
    the largest region (:376-391), and report area / volume (:452-458).  Returns ``(verts, faces, measures)``."""
    def zkey(name):
        m = re.search(r"_(\d+)_(\d+)\.vtp$", name)
        return int(m.group(1)) if m else 0

    soups = []
    for name in sorted(filenames, key=zkey):
        v, f = read_vtp(name)
        if len(f):
            soups.append(v[f])
    if not soups:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), {"volume": 0.0, "area": 0.0}
    soup = np.concatenate(soups).reshape(-1, 3)
    key = np.ascontiguousarray(soup).view([("", np.uint32)] * 3).ravel()
    _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # vertices numbered by first appearance
    rank = np.empty(len(order), np.int64)
    rank[order] = np.arange(len(order))
    verts = soup[first[order]]
    faces = rank[inverse].reshape(-1, 3).astype(np.int32)
    if keep_largest_region and len(faces):
        verts, faces, _ = keep_largest(verts, faces)
    volume, area = mass_properties(verts, faces)
    return verts, faces, {"volume": volume, "area": area}


_keep_largest_region = keep_largest  # (join_process_surface has a parameter of that name, like the reference)


def fill_holes(verts, faces, hole_size=300.0):
    """The hole-filling step of join_process_surface (vtkFillHolesFilter with SetHoleSize(300),
    invesalius/data/surface_process.py:396-416) on the GPU (csrc/k_meshtail.hip): every closed rim of boundary edges whose
    bounding sphere (half the diagonal of the rim's bounding box) has a radius <= `hole_size` is closed with new triangles,
    appended after the old ones and wound so that the patch continues the surface's orientation.  VTK triangulates the rim's
    polygon without new points; here the rim is fanned from its centroid (ONE new point per hole), which closes any rim --
    planar or not, convex or not -- without a geometric predicate.  Rims that touch in one vertex (pinch points) are walked
    apart, open chains of a non-manifold rim are left alone.  PARITY UNPINNED: VTK is third party and not installed; the tests
    pin that the result is closed and consistently oriented, and compare the kernels array for array with the same rules in
    plain Python (tests/_mesh_tail_ref.py).  Returns (verts, faces, number of holes filled)."""
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(np.asarray(faces, np.int32).reshape(-1, 3))
    if not len(f):
        return v, f, 0
    # ONE call (the size-query-then-fill protocol ran the whole pipeline -- upload, edge hash, rim walk, scans -- twice): room for
    # the worst case, one new point per three rim edges and one cap triangle per rim edge, of which every edge of the mesh could be
    # one; np.empty pages that are never written cost nothing
    cap_v, cap_t = len(f), 3 * len(f)
    while True:
        try:
            new_v, new_f = np.empty((cap_v, 3), np.float32), np.empty((cap_t, 3), np.int32)
        except MemoryError:  # a host that commits at allocation time (vm.overcommit_memory=2): start small, grow on demand
            cap_v, cap_t = max(1024, cap_v // 16), max(3072, cap_t // 16)
            continue
        nv, nt = ctypes.c_int64(cap_v), ctypes.c_int64(cap_t)
        rc = L.lib().ivx_mesh_fill_holes(L.ptr(v), ctypes.c_int64(len(v)), L.ptr(f), ctypes.c_int64(len(f)), ctypes.c_double(float(hole_size)),
                                         L.ptr(new_v), L.ptr(new_f), ctypes.byref(nv), ctypes.byref(nt))
        if rc == L.IVX_EINVAL and cap_t < 3 * len(f) and "room for" in L.last_error():  # the smaller buffers of the fallback were too small
            cap_v, cap_t = min(len(f), cap_v * 4), min(3 * len(f), cap_t * 4)
            continue
        L.check(rc, "mesh_fill_holes")
        break
    if nv.value == 0:
        return v, f, 0
    # (np.concatenate copies: nothing returned keeps the worst-case buffers alive)
    return np.concatenate([v, new_v[:nv.value]]), np.concatenate([f, new_f[:nt.value]]), int(nv.value)


def point_normals(verts, faces, feature_angle=80.0, splitting=True, auto_orient=True):
    """The last filter of join_process_surface (vtkPolyDataNormals: FeatureAngle 80, SplittingOn, AutoOrientNormalsOn,
    ComputeCellNormalsOn, invesalius/data/surface_process.py:420-435) on the GPU (csrc/k_meshtail.hip): unit cell normals;
    points on an edge sharper than the feature angle are duplicated, one copy per fan of triangles joined by smooth edges (the
    fan of the vertex's smallest corner keeps the point, the copies follow the old points in (vertex, fan) order); a point's
    normal is the normalised sum of its fan's unit cell normals; with auto-orientation the (consistently wound) surface is
    turned so that its normals point out of the enclosed volume.  PARITY UNPINNED (VTK absent): pinned by properties -- unit
    length, no copy without a sharp edge, a cube gets 24 points, a smooth sphere none extra, normals of a closed surface point
    outwards -- and array for array against the same rules in plain Python (tests/_mesh_tail_ref.py).
    Returns (verts, faces, point normals float32, cell normals float32)."""
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(np.asarray(faces, np.int32).reshape(-1, 3))
    if not len(f):
        return v, f, np.zeros((len(v), 3), np.float32), np.zeros((0, 3), np.float32)
    lib = L.lib()
    cosang = ctypes.c_double(float(np.cos(np.deg2rad(feature_angle))))
    # ONE call with room for the worst case (every corner its own point): pages of np.empty that are never written cost nothing
    cap = len(v) + 3 * len(f)
    args = (L.ptr(v), ctypes.c_int64(len(v)), L.ptr(f), ctypes.c_int64(len(f)), cosang, int(bool(splitting)), int(bool(auto_orient)))
    try:
        out_v, pn = np.empty((cap, 3), np.float32), np.empty((cap, 3), np.float32)
    except MemoryError:  # a host that commits at allocation time: ask for the size first (the two-call protocol)
        n = ctypes.c_int64(0)
        L.check(lib.ivx_mesh_point_normals(*args, None, None, None, None, ctypes.byref(n)), "mesh_point_normals")
        cap = int(n.value)
        out_v, pn = np.empty((cap, 3), np.float32), np.empty((cap, 3), np.float32)
    n = ctypes.c_int64(cap)
    out_f, cn = np.empty((len(f), 3), np.int32), np.empty((len(f), 3), np.float32)
    L.check(lib.ivx_mesh_point_normals(*args, L.ptr(out_v), L.ptr(out_f), L.ptr(pn), L.ptr(cn), ctypes.byref(n)), "mesh_point_normals")
    # shrink in place (no copy, and no view that keeps the worst-case buffers alive)
    out_v.resize((n.value, 3), refcheck=False)
    pn.resize((n.value, 3), refcheck=False)
    return out_v, out_f, pn, cn


def join_process_surface(filenames, algorithm, smooth_iterations, smooth_relaxation_factor, decimate_reduction, keep_largest,
                         fill_holes, options, msg_queue=None):
    """``invesalius.data.surface_process.join_process_surface`` (:204-472) with its own signature and return value
    ``(filename_of_the_full_vtp, {"volume": v, "area": a})``, on the piece files `create_surface_piece` wrote:

    * append + clean (vtkAppendPolyData, vtkCleanPolyData with point merging, :228-268): `join_surface_pieces`;
    * ``algorithm == "ca_smoothing"`` (:270-322): cell normals and the context-aware smoothing on the GPU with
      ``options["angle" | "max distance" | "min weight" | "steps"]``;
    * decimation (:350-373): the reference's condition is inverted (``if not decimate_reduction``), so vtkQuadricDecimation
      only ever runs with a target reduction of 0, i.e. changes nothing -- nothing to do here either;
    * ``keep_largest`` (:378-391) and the area / volume of :452-458 on the GPU;
    * ``fill_holes`` (:396-416, vtkFillHolesFilter, hole size 300): `fill_holes` below -- surfaces made with
      ``fill_border_holes`` are closed and pass through unchanged, open rims up to the hole size get a cap;
    * the final vtkPolyDataNormals (:420-435: feature angle 80, splitting, auto-orientation, cell normals): `point_normals`
      below; the file carries the split points, the triangles and both normal arrays; area and volume are those of the
      surface before the split, like the reference's ``to_measure``.  Both are restatements of VTK filters that are not
      installed here: parity unpinned, pinned by properties (tests/test_host_logic.py).
    The reference's progress messages go to `msg_queue` unchanged."""
    import queue as _queue

    from . import invesalius_rs as rs

    def send_message(msg):
        if msg_queue is None:
            return
        try:
            msg_queue.put_nowait(msg)
        except _queue.Full as e:
            print(e)

    send_message("Joining surfaces ...")
    send_message("Cleaning surface ...")
    verts, faces, _ = join_surface_pieces(list(filenames))
    if algorithm == "ca_smoothing" and len(faces):
        send_message("Calculating normals ...")
        send_message("Context Aware smoothing ...")
        mesh = rs.Mesh.from_indexed(np.ascontiguousarray(verts, np.float32), faces)  # unit cell normals (:271-285)
        rs.ca_smoothing(mesh, options["angle"], options["max distance"], options["min weight"], options["steps"])
        verts = np.asarray(mesh.vertices, np.float32)
    if not decimate_reduction:
        send_message("Decimating ...")  # (target reduction 0: see above)
    if keep_largest and len(faces):
        send_message("Finding the largest ...")
        verts, faces, _ = _keep_largest_region(verts, faces)
    if fill_holes and len(faces):
        send_message("Filling holes ...")
        verts, faces, _ = globals()["fill_holes"](verts, faces, 300.0)  # (the parameter shadows the function, as in the reference)
    send_message("Calculating area and volume ...")
    volume, area = mass_properties(verts, faces) if len(faces) else (0.0, 0.0)  # (:452-458: measured BEFORE the points are split)
    nverts, nfaces, pn, cn = point_normals(verts, faces, 80.0, True, True)
    fd, filename = tempfile.mkstemp(suffix="_full.vtp")
    os.close(fd)
    write_vtp(filename, nverts, nfaces, pn, cn)
    return filename, {"volume": float(volume), "area": float(area)}


def resize_image_array(image, resolution_percentage, as_mmap=False):
    """``imagedata_utils.resize_image_array`` (:121-130): ``scipy.ndimage.zoom(image, factor, image.dtype, order=2)``, the
    down-sampling AddNewActor applies to image AND mask for the Low / Medium quality presets (surface.py:1350-1353), on
    the GPU (csrc/k_zoom.hip: quadratic B-spline prefilter + interpolation in float64, scipy's operation order)."""
    a = np.asarray(image)
    if a.ndim != 3 or a.dtype not in (np.int16, np.uint8):
        raise TypeError("resize_image_array: 3-D int16 or uint8 volume expected")
    a = np.ascontiguousarray(a)
    oshape = tuple(int(round(s * float(resolution_percentage))) for s in a.shape)
    out = np.empty(oshape, a.dtype)
    L.check(L.lib().ivx_zoom_order2(L.DT[a.dtype], L.ptr(a), L.i64(a.shape), L.ptr(out), L.i64(oshape)), "zoom")
    if as_mmap:
        fd, fname = tempfile.mkstemp(suffix="_resized")
        os.close(fd)
        mm = np.memmap(fname, shape=out.shape, dtype=out.dtype, mode="w+")
        mm[:] = out
        return mm
    return out


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
        parts.append(surface_piece(image, mask_matrix, roi, spacing, min_value, max_value, from_binary,
                                   fill_border_holes))
    return np.concatenate(parts) if parts else np.empty((0, 3, 3), np.float32)


def join_process_volume(image, mask_matrix, spacing, min_value, max_value, from_binary, fill_border_holes=True,
                        keep_largest_region=False):
    """The geometry of join_process_surface (surface_process.py:204-472) straight from the resident arrays: the pieces'
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
