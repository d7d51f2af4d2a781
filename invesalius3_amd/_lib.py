"""ctypes binding of libivx.so (the C ABI declared in include/ivx.h).

This is the stub a reference maintainer would add in place of ``from invesalius_rs import _native``
(invesalius_rs/__init__.py:8).  There is no CPU fallback: if the shared library is missing the import of any
compute entry point raises, and on a machine without a HIP device every call returns IVX_EHIP -> RuntimeError.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IVX_LIB_PATH") or os.path.join(_HERE, "libivx.so")  # IVX_LIB_PATH: A/B builds of the same ABI

IVX_OK, IVX_EINVAL, IVX_ERANGE, IVX_ENOMEM, IVX_EDOM, IVX_EHIP = 0, -1, -2, -3, -4, -5
U8, I16, F64, U16, F32, I32, I64, I8 = 0, 1, 2, 3, 4, 5, 6, 7
DT = {np.dtype(np.uint8): U8, np.dtype(np.int16): I16, np.dtype(np.float64): F64, np.dtype(np.uint16): U16, np.dtype(np.float32): F32, np.dtype(np.int32): I32,
      np.dtype(np.int64): I64}
MIP_MAX, MIP_MIN, MIP_MEAN, MIP_SUM = 0, 1, 2, 3

c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p


class McParams(ctypes.Structure):
    """struct ivx_mc_params (include/ivx.h)."""
    _fields_ = [
        ("dtype", ctypes.c_int32), ("pad_xy", ctypes.c_int32), ("pad_bottom", ctypes.c_int32),
        ("pad_top", ctypes.c_int32), ("vtk_pz", ctypes.c_int32), ("niso", ctypes.c_int32),
        ("nz", c_i64), ("ny", c_i64), ("nx", c_i64), ("roi_start", c_i64),
        ("pad_value", ctypes.c_double), ("spacing", ctypes.c_double * 3), ("iso", ctypes.c_double * 2),
    ]


class FloodPlan(ctypes.Structure):
    """struct ivx_flood_plan (include/ivx.h)."""
    _fields_ = [("dz", c_i64), ("dy", c_i64), ("dx", c_i64), ("wx", c_i64), ("strct_bits", ctypes.c_uint32)]


_lib = None


def lib() -> ctypes.CDLL:
    """Load libivx.so (built in-tree by invesalius3_amd/build.py).  Fails loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libivx.so not found at %s: build it with `python -m invesalius3_amd.build` "
                "(there is no CPU fallback in this package)" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ivx_last_error.restype = ctypes.c_char_p
    return _lib


def last_error() -> str:
    return (lib().ivx_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "ivx") -> int:
    """Map the C status to the exception the reference raises for the same condition."""
    if rc >= 0:
        return rc
    msg = "%s: %s" % (what, last_error())
    if rc == IVX_EINVAL:
        raise TypeError(msg)
    if rc == IVX_ERANGE:
        raise IndexError(msg)
    if rc == IVX_ENOMEM:
        raise MemoryError(msg)
    if rc == IVX_EDOM:
        raise ValueError(msg)
    raise RuntimeError(msg)


def i64(seq):
    return (c_i64 * len(seq))(*[int(v) for v in seq])


def ptr(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


def dtype_code(a: np.ndarray, allowed=(U8, I16, F64, U16)) -> int:
    code = DT.get(a.dtype)
    if code is None or code not in allowed:
        raise TypeError("unsupported array dtype %s" % a.dtype)
    return code


class _Pinned:
    """owner of one hipHostMalloc block (freed when the last numpy view of it goes away)"""

    def __init__(self, nbytes: int):
        self.ptr = ctypes.c_void_p()
        check(lib().ivx_host_alloc(ctypes.byref(self.ptr), ctypes.c_size_t(int(nbytes))), "ivx_host_alloc")

    def __del__(self):
        try:
            if self.ptr and self.ptr.value:
                lib().ivx_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """np.empty in page-locked host memory: uploads from it and downloads into it run at the PCIe link's rate (a pageable
    array goes through the runtime's bounce buffers).  For callers that can keep their volume / results there."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    own = _Pinned(max(n, 16))
    buf = (ctypes.c_uint8 * max(n, 16)).from_address(own.ptr.value)
    buf._ivx_owner = own  # the ctypes array keeps the block alive, numpy keeps the ctypes array alive
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def device_count() -> int:
    n = ctypes.c_int(0)
    lib().ivx_device_count(ctypes.byref(n))
    return n.value


def require_device():
    if device_count() < 1:
        raise RuntimeError("invesalius3_amd: no HIP device visible and there is no CPU fallback")


def device_name() -> str:
    buf = ctypes.create_string_buffer(256)
    check(lib().ivx_device_name(buf, ctypes.c_size_t(256)))
    return buf.value.decode()


def set_device(i: int):
    check(lib().ivx_set_device(int(i)))


def synchronize():
    check(lib().ivx_device_synchronize())
