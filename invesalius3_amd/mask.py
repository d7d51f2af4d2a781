"""Host mirror of the mask operations of invesalius/data/mask.py that sit on the voxel path.

`fill_holes_auto` is Mask.fill_holes_auto (mask.py:519-562) without the undo history: the reference labels the
inverted mask with scipy.ndimage.label on the host (the dominant cost) and hands the labels to the Rust function; here
labelling and filling are one device pass over bit planes (csrc/k_holes.hip), no label volume is ever built."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib as L

CON2D = {4: 1, 8: 2}
CON3D = {6: 1, 18: 2, 26: 3}


def _structure(rank: int, connectivity: int) -> np.ndarray:
    """scipy.ndimage.generate_binary_structure(rank, connectivity): offsets with at most `connectivity` non-zero axes."""
    grid = np.indices((3,) * rank) - 1
    return (np.abs(grid).sum(axis=0) <= connectivity).astype(np.uint8)


def fill_holes_auto(matrix: np.ndarray, target: str, conn: int, orientation: str, index: int, size: int) -> bool:
    """``matrix`` is the padded (dz+1, dy+1, dx+1) uint8 mask matrix, edited in place.  target "3D": holes of the whole
    volume under 6/18/26-connectivity; otherwise the holes of slice ``index`` of ``orientation`` ("AXIAL", "CORONAL",
    "SAGITAL") under 4/8-connectivity.  Returns whether anything changed (what decides the reference's
    save_history)."""
    if matrix.dtype != np.uint8 or matrix.ndim != 3:
        raise TypeError("mask matrix must be a 3-D uint8 array")
    if target == "3D":
        view = matrix[1:, 1:, 1:]
        strct = _structure(3, CON3D[conn])
        shape, strides = view.shape, view.strides
    else:
        if orientation == "AXIAL":
            view = matrix[index + 1, 1:, 1:]
        elif orientation == "CORONAL":
            view = matrix[1:, index + 1, 1:]
        elif orientation == "SAGITAL":
            view = matrix[1:, 1:, index + 1]
        else:
            raise ValueError("orientation must be AXIAL, CORONAL or SAGITAL")
        strct = _structure(2, CON2D[conn]).reshape(1, 3, 3)
        shape, strides = (1,) + view.shape, (0,) + view.strides
    if int(size) < 0:
        raise OverflowError("size must be non-negative")
    strct = np.ascontiguousarray(strct)
    sshape = strct.shape if strct.ndim == 3 else (1,) + strct.shape
    modified = ctypes.c_int(0)
    L.check(L.lib().ivx_fill_holes_auto(L.ptr(view), L.i64(shape), L.i64(strides), L.ptr(strct), L.i64(sshape),
                                        ctypes.c_uint32(min(int(size), 0xFFFFFFFF)), ctypes.byref(modified)),
            "fill_holes_auto")
    return bool(modified.value)
