"""Host mirror of the array work behind the 3-D region-growing tool,
FloodFillSegmentInteractorStyle.do_3d_seg + do_rg_confidence (invesalius/data/styles.py:3151-3251): everything between
the mouse click and `save_history`, on the GPU.  The GUI parts (picker, progress dialog, undo history) stay with the
caller; the function takes what they produce -- the clicked voxel and the dialog's configuration -- and edits the mask
matrix in place."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib as L
from . import slice_ as sl
from .device import DeviceVolume
from .mask import CON3D, _structure


def do_3d_seg(image: np.ndarray, mask_matrix: np.ndarray, seed_xyz, method: str = "dynamic", con_3d: int = 6,
              fill_value: int = 254, t0=None, t1=None, dev_min=25, dev_max=25, use_ww_wl: bool = False, ww=None, wl=None,
              confid_mult: float = 2.5, confid_iters: int = 3, threshold_range=None) -> bool:
    """``image`` (dz,dy,dx) int16, ``mask_matrix`` the padded uint8 matrix of the current mask, ``seed_xyz`` the picked
    voxel.  method "threshold" (t0, t1 given), "dynamic" (seed value -dev_min / +dev_max, optionally on the
    get_LUT_value_255 image) or "confidence".  ``threshold_range`` = the mask's threshold range, for the
    do_threshold_to_all_slices() the reference runs first (slices whose flag is 0 are (re)thresholded).
    Returns False when the click is rejected (seed outside [t0, t1], styles.py:3178), True after editing the mask."""
    if image.dtype != np.int16 or image.ndim != 3:
        raise TypeError("image must be a 3-D int16 array")
    if mask_matrix.dtype != np.uint8 or mask_matrix.shape != tuple(s + 1 for s in image.shape):
        raise TypeError("mask matrix must be uint8 and one voxel larger than the image on every axis")
    if method not in ("threshold", "dynamic", "confidence"):
        raise ValueError("method must be threshold, dynamic or confidence")
    x, y, z = (int(v) for v in seed_xyz)
    if not (0 <= z < image.shape[0] and 0 <= y < image.shape[1] and 0 <= x < image.shape[2]):
        raise IndexError("seed outside the volume")
    strct = _structure(3, CON3D[con_3d])
    with DeviceVolume(np.ascontiguousarray(image)) as vol:
        flood_image = None
        if use_ww_wl and method in ("dynamic", "confidence"):
            flood_image = vol.lut_image_255(ww, wl)  # get_LUT_value_255(image, ww, wl), int16
        if method != "confidence":
            if method == "threshold":
                lo, hi = t0, t1
                v = int(image[z, y, x])
            else:
                if flood_image is not None:
                    one = np.zeros(1, np.int16)
                    L.check(L.lib().ivx_memcpy_d2h(L.ptr(one), flood_image.at(((z * image.shape[1] + y) * image.shape[2] + x) * 2),
                                                   ctypes.c_size_t(2)))
                    v = int(one[0])
                else:
                    v = int(image[z, y, x])
                lo, hi = v - dev_min, v + dev_max
            if v < lo or v > hi:
                if flood_image is not None:
                    flood_image.close()
                return False
        if threshold_range is not None:
            sl.do_threshold_to_all_slices(mask_matrix, image, threshold_range)
        vol.mask.upload(np.ascontiguousarray(mask_matrix[1:, 1:, 1:]))
        vol.zero_out_mask()
        if method == "confidence":
            vol.region_grow_confidence((x, y, z), strct, confid_mult, confid_iters, select_value=int(fill_value),
                                       image=flood_image)
        else:
            vol.region_grow([(x, y, z)], lo, hi, strct, fill=1, select_value=int(fill_value), image=flood_image)
        mask_matrix[1:, 1:, 1:] = vol.download_mask()
        if flood_image is not None:
            flood_image.close()
    return True


CON2D = {4: 1, 8: 2}


def flood_fill_mask(mask_matrix: np.ndarray, seed_xyz, target: str = "3D", orientation: str = "AXIAL", con_2d: int = 4,
                    con_3d: int = 6, t0: int = 0, t1: int = 2, fill_value: int = 254, image: np.ndarray | None = None,
                    threshold_range=None) -> bool:
    """The array work of FloodFillMaskInteractorStyle.OnFFClick (invesalius/data/styles.py:2477-2569) -- the "fill holes" tool
    (t0..t1 = 0..2 -> 254) and, with ``t0=253, t1=255, fill_value=1``, RemoveMaskPartsInteractorStyle (:2572-2589): the
    clicked voxel must hold a value in [t0, t1] (else nothing happens: False); "3D" brings the stale slices up to date first
    (`image` and `threshold_range` of the mask) and floods with the 6 / 18 / 26 structure, "2D" floods inside the clicked
    slice only, with the 4 / 8 structure laid into the slice's plane (:2507-2515).  Edits `mask_matrix` in place."""
    from . import invesalius_rs as floodfill

    if mask_matrix.dtype != np.uint8 or mask_matrix.ndim != 3:
        raise TypeError("mask matrix must be a 3-D uint8 array")
    x, y, z = (int(v) for v in seed_xyz)
    mask = mask_matrix[1:, 1:, 1:]
    if mask[z, y, x] < t0 or mask[z, y, x] > t1:
        return False
    if target == "3D":
        bstruct = _structure(3, CON3D[con_3d])
        if threshold_range is not None and image is not None:
            sl.do_threshold_to_all_slices(mask_matrix, image, threshold_range)
    else:
        b2 = _structure(2, CON2D[con_2d])
        shape, where = {"AXIAL": ((1, 3, 3), (0, slice(None), slice(None))), "CORONAL": ((3, 1, 3), (slice(None), 0, slice(None))),
                        "SAGITAL": ((3, 3, 1), (slice(None), slice(None), 0))}[orientation]
        bstruct = np.zeros(shape, dtype="uint8")
        bstruct[where] = b2
    floodfill.floodfill_threshold_inplace(mask, ((x, y, z),), t0, t1, fill_value, bstruct)
    return True


def select_mask_part(mask_matrix: np.ndarray, select_matrix: np.ndarray, seed_xyz, con_3d: int = 6, remove: bool = False,
                     image: np.ndarray | None = None, threshold_range=None):
    """The array work of SelectMaskPartsInteractorStyle.OnSelect (invesalius/data/styles.py:2883-2960): after the threshold of
    the stale slices, a click copies the connected part of `mask_matrix` (values 253..255) that holds the clicked voxel into
    the selection mask as 254; a Ctrl+click (`remove`) clears the connected part of the SELECTION (254..255 -> 0) that holds
    it.  Both matrices are the padded (dz+1, dy+1, dx+1) uint8 matrices; `select_matrix` is edited in place."""
    from . import invesalius_rs as floodfill

    x, y, z = (int(v) for v in seed_xyz)
    shape = mask_matrix.shape
    if x < 0 or y < 0 or z < 0 or z >= shape[0] - 1 or y >= shape[1] - 1 or x >= shape[2] - 1:
        return
    bstruct = _structure(3, CON3D[con_3d])
    if threshold_range is not None and image is not None:
        sl.do_threshold_to_all_slices(mask_matrix, image, threshold_range)
    sel = select_matrix[1:, 1:, 1:]
    if remove:
        floodfill.floodfill_threshold(sel, ((x, y, z),), 254, 255, 0, bstruct, sel)
    else:
        floodfill.floodfill_threshold(mask_matrix[1:, 1:, 1:], ((x, y, z),), 253, 255, 254, bstruct, sel)


BRUSH_FOREGROUND, BRUSH_BACKGROUND = 1, 2  # invesalius/constants.py


def watershed_brush_release(image_matrix: np.ndarray, mask_matrix: np.ndarray, markers_matrix: np.ndarray, n: int,
                            orientation: str = "AXIAL", algorithm: str = "Watershed", con_2d: int = 4, mg_size=3,
                            use_ww_wl: bool = True, wl=0, ww=0, overwrite: bool = False) -> bool:
    """The array work of WaterShedInteractorStyle.OnBrushRelease (invesalius/data/styles.py:1926-1997), the 2-D watershed of the
    slice the brush was released on: slice `n` of the image, of the markers and of the padded mask matrix along `orientation`
    (the AXIAL branch also sets the slice's "thresholded" flag, :1932), nothing unless both brush values are present, then the
    cost / gradient image, the flood with the 4 / 8 structure and the merge rule of :1984-1989, in place on the mask.
    The reference's IFT branch without window/level computes ``image - image.min().astype("uint16")`` -- not an unsigned image
    -- and scipy refuses it: TypeError, here as there."""
    from . import watershed_process as wp

    if orientation == "AXIAL":
        image, mask, markers = image_matrix[n], mask_matrix[n + 1, 1:, 1:], markers_matrix[n]
        mask_matrix[n + 1, 0, 0] = 1
    elif orientation == "CORONAL":
        image, mask, markers = image_matrix[:, n, :], mask_matrix[1:, n + 1, 1:], markers_matrix[:, n, :]
    elif orientation == "SAGITAL":
        image, mask, markers = image_matrix[:, :, n], mask_matrix[1:, 1:, n + 1], markers_matrix[:, :, n]
    else:
        raise ValueError("orientation must be AXIAL, CORONAL or SAGITAL")
    if not ((markers == BRUSH_BACKGROUND).any() and (markers == BRUSH_FOREGROUND).any()):
        return False
    bstruct = _structure(2, CON2D[con_2d])
    image = np.ascontiguousarray(image)
    mk = np.ascontiguousarray(markers).astype("int16")
    if algorithm == "Watershed":
        tmp_mask = wp.watershed(wp.cost_image(image, use_ww_wl, wl, ww, mg_size), mk, bstruct)
    elif use_ww_wl:
        tmp_mask = wp.watershed_ift(wp.cost_image(image, True, wl, ww, 0), mk, bstruct)
    else:
        raise TypeError("only 8 and 16 unsigned inputs are supported")  # scipy's message for the reference's int image (:1975-1976)
    tmp = np.ascontiguousarray(tmp_mask.astype(np.uint8))
    m2 = np.ascontiguousarray(mask)
    wp.merge(m2, tmp, bool(overwrite))
    mask[...] = m2
    return True
