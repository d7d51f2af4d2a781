"""Headless driver: a .inv3 project in, mask / surface / measurements out, every voxel and triangle stage on the GPU.

    python -m invesalius3_amd.headless CASE.inv3 --threshold 226 3071 --seed 250 260 100 --largest --smooth \\
        --stl bone.stl --save CASE_out.inv3

What the reference does through its GUI for the same result: Slice.SetMaskThreshold / do_threshold_to_all_slices
(invesalius/data/slice_.py:1240-1247, 1739-1769), the region-growing tool (styles.py:3151-3216),
SurfaceManager.AddNewActor -> create_surface_piece / join_process_surface (surface.py:1362-1380,
surface_process.py:71-472) and vtkSTLWriter (surface.py:1827-1829).  No wx, no VTK here; one JSON line on stdout."""
from __future__ import annotations

import argparse
import ctypes
import json
import sys
import time

import numpy as np

from . import _lib as L
from . import project as prj
from . import surface_process as sp
from .device import DeviceVolume, c64


def run(args) -> dict:
    t_all = time.perf_counter()
    proj = prj.open_inv3(args.project)
    out = {"project": proj.name, "shape": list(proj.matrix_shape), "spacing": list(proj.spacing), "dtype": proj.matrix_dtype}
    if proj.matrix.dtype != np.int16:
        raise TypeError("the GPU path takes int16 volumes (found %s)" % proj.matrix.dtype)
    vol = DeviceVolume(np.ascontiguousarray(proj.matrix), spacing=proj.spacing)
    lib = L.lib()
    try:
        if args.threshold is not None:
            lo, hi = args.threshold
            with vol.timer.span("threshold"):
                vol.threshold(lo, hi)
            out["threshold"] = [lo, hi]
        else:
            if args.mask not in proj.masks:
                raise KeyError("project holds no mask %d (masks: %s)" % (args.mask, sorted(proj.masks)))
            rec = proj.masks[args.mask]
            vol.mask.upload(np.ascontiguousarray(rec.interior))
            lo, hi = rec.threshold_range
            out["mask"] = {"index": args.mask, "name": rec.name, "threshold_range": [lo, hi]}
        if args.seed:
            if args.threshold is None and rec.edited:
                # the region is grown in the IMAGE inside the mask's threshold range; an edited mask (brush, cut, earlier
                # region growing) is no longer that threshold, and growing would silently throw the edits away
                raise SystemExit("--seed with --mask %d: that mask was edited by hand; region growing floods the image inside "
                                 "the mask's threshold range and would discard the edits -- use --threshold LO HI instead"
                                 % args.mask)
            seeds = [tuple(args.seed[i:i + 3]) for i in range(0, len(args.seed), 3)]
            strct = np.ones((3, 3, 3), np.uint8) if args.connectivity == 26 else _strct(args.connectivity)
            vol.zero_out_mask()
            with vol.timer.span("region_grow"):
                rounds = vol.region_grow(seeds, lo, hi, strct, fill=1, select_value=None)
            vol.mask.zero(vol.stream)  # keep only the grown region
            L.check(lib.ivx_dev_flood_apply_where(vol.mask.ptr, vol.out_mask.ptr, c64(vol.n), 1, 255, vol.stream))
            out["region_grow"] = {"seeds": [list(s) for s in seeds], "rounds": rounds, "voxels": vol.reached_count()}
        mask = vol.download_mask()
        out["mask_voxels"] = int(np.count_nonzero(mask >= 127))
        with vol.timer.span("surface"):
            nv, nt = vol.marching_cubes_indexed(from_binary=True, fill_border_holes=True)
        verts_buf, faces_buf = vol._verts, vol._faces
        out["surface"] = {"vertices": nv, "triangles": nt}
        keep_v = keep_f = None
        if args.largest and nt:
            from .device import DeviceBuffer
            keep_v, keep_f = DeviceBuffer(nv * 12 + 16), DeviceBuffer(nt * 12 + 16)
            n1, n2, nr = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
            with vol.timer.span("keep_largest"):
                L.check(lib.ivx_dev_mesh_keep_largest(verts_buf.ptr, c64(nv), faces_buf.ptr, c64(nt), keep_v.ptr, c64(nv),
                                                      keep_f.ptr, c64(nt), ctypes.byref(n1), ctypes.byref(n2),
                                                      ctypes.byref(nr), vol.stream), "keep_largest")
            verts_buf, faces_buf, nv, nt = keep_v, keep_f, n1.value, n2.value
            out["largest"] = {"regions": nr.value, "vertices": nv, "triangles": nt}
        if args.smooth and nt:
            from .device import DeviceBuffer
            nrm = DeviceBuffer(nt * 24 + 16)
            with vol.timer.span("smooth"):
                L.check(lib.ivx_dev_mesh_face_normals(verts_buf.ptr, L.F32, faces_buf.ptr, c64(nt), nrm.ptr, vol.stream))
                L.check(lib.ivx_dev_context_aware_smoothing(verts_buf.ptr, L.F32, c64(nv), faces_buf.ptr, c64(nt), nrm.ptr,
                                                            ctypes.c_double(args.angle), ctypes.c_double(args.max_distance),
                                                            ctypes.c_double(args.min_weight), ctypes.c_int(args.steps), None,
                                                            None, vol.stream), "ca_smoothing")
            nrm.close()
            out["smooth"] = {"angle": args.angle, "max_distance": args.max_distance, "min_weight": args.min_weight,
                             "steps": args.steps}
        from .device import DeviceBuffer
        mass = DeviceBuffer(64)
        with vol.timer.span("mass"):
            L.check(lib.ivx_dev_mesh_mass_properties(verts_buf.ptr, faces_buf.ptr, c64(nt), mass.ptr, vol.stream))
        vol.sync()
        m = mass.download((8,), np.float64)
        out["volume"], out["area"] = float(m[0]), float(m[1])
        mass.close()
        if args.stl:
            verts = verts_buf.download((nv, 3), np.float32)
            faces = faces_buf.download((nt, 3), np.int32)
            sp.write_stl_binary(args.stl, verts[faces])
            out["stl"] = args.stl
        if args.save:
            rec = prj.new_mask(proj, args.mask_name, (lo, hi))
            rec.matrix[1:, 1:, 1:] = mask
            rec.matrix[1:, 0, 0] = 1  # per-slice "already thresholded" flags, as SetMaskThreshold leaves them (slice_.py:1246)
            prj.save_inv3(args.save, proj)
            out["saved"] = args.save
        out["gpu_ms"] = {k: round(float(sum(v)), 4) for k, v in vol.timer.collect().items()}
        for b in (keep_v, keep_f):
            if b is not None:
                b.close()
    finally:
        vol.close()
        proj.close()
    out["wall_s"] = round(time.perf_counter() - t_all, 3)
    return out


def _strct(conn: int) -> np.ndarray:
    from .mask import CON3D, _structure
    return _structure(3, CON3D[conn])


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m invesalius3_amd.headless", description=__doc__.split("\n")[0])
    ap.add_argument("project", help=".inv3 file")
    g = ap.add_mutually_exclusive_group()
    g.add_argument("--threshold", nargs=2, type=int, metavar=("LO", "HI"), help="threshold the image into a new mask")
    g.add_argument("--mask", type=int, default=0, help="use the project's mask with this index (default 0)")
    ap.add_argument("--seed", nargs="+", type=int, default=None, metavar="X Y Z", help="keep the region grown (in the image, inside the threshold range) from these voxels; refused for a hand-edited --mask")
    ap.add_argument("--connectivity", type=int, choices=(6, 18, 26), default=26)
    ap.add_argument("--largest", action="store_true", help="keep the largest connected surface")
    ap.add_argument("--smooth", action="store_true", help="context-aware smoothing")
    ap.add_argument("--angle", type=float, default=0.7)
    ap.add_argument("--max-distance", type=float, default=3.0)
    ap.add_argument("--min-weight", type=float, default=0.5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--stl", help="write the surface as binary STL")
    ap.add_argument("--save", help="write the project back with the new mask appended")
    ap.add_argument("--mask-name", default="GPU mask")
    args = ap.parse_args(argv)
    if args.seed and len(args.seed) % 3:
        ap.error("--seed takes triples of x y z")
    print(json.dumps(run(args)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
