"""The Z-slab communicator: ctypes veneer over the ``ivx_comm_*`` C ABI (csrc/ivx_comm.hip = RCCL over xGMI).

One process per GPU.  Every collective takes DEVICE pointers and only enqueues on a HIP stream; the conveniences for
small host arrays (projection images, ranges, flags) stage through one cached device buffer.

The pointer-level protocol (``exchange``, ``exchange_vote``, ``allreduce``, ``allgather``, ``bcast``, ``send``, ``recv``,
``stage``, ``sync``) is what ``invesalius3_amd.parallel`` is written against; the tests drive the same orchestration
with stand-ins that implement it over host memory (tests/_ptr_comm.py: gloo, and an in-process loop-back on one GPU).

Rendezvous without a launcher library: rank 0 makes the 128-byte RCCL id and writes it to a file every rank can see
(``IVX_COMM_FILE``, or a name derived from the launcher's pid and MASTER_PORT under ``torchrun``); the others wait for it.
"""
from __future__ import annotations

import ctypes
import os
import time

import numpy as np

from . import _lib as L

OP = {"sum": 0, "max": 1, "min": 2}
_WIDE = {np.dtype(np.int16): np.int32, np.dtype(np.uint16): np.int32, np.dtype(np.uint8): np.int32,
         np.dtype(np.int8): np.int32, np.dtype(np.bool_): np.int32, np.dtype(np.float32): np.float32,
         np.dtype(np.float64): np.float64, np.dtype(np.int32): np.int32, np.dtype(np.int64): np.int64}
_CODE = {np.dtype(np.int32): L.I32, np.dtype(np.int64): L.I64, np.dtype(np.float32): L.F32, np.dtype(np.float64): L.F64}


def _vp(p):
    if p is None:
        return None
    return p if isinstance(p, ctypes.c_void_p) else ctypes.c_void_p(int(p))


class HostArrayOps:
    """Host-array conveniences on top of the pointer-level protocol (mixed into every communicator)."""

    def allreduce_array(self, a: np.ndarray, op: str) -> np.ndarray:
        """Element-wise sum / max / min of one small host array per rank (RCCL reduces no 16-bit integers: widened)."""
        a = np.ascontiguousarray(a)
        wide = a.astype(_WIDE[a.dtype])
        if self.world == 1:
            return a.copy()
        s = self.stage(wide.nbytes)
        s.upload(wide)
        self.allreduce(s.ptr, wide.size, _CODE[wide.dtype], OP[op], None)
        self.sync()
        return s.download(wide.shape, wide.dtype).astype(a.dtype)

    def allreduce_sum(self, value: int) -> int:
        return int(self.allreduce_array(np.array([int(value)], np.int64), "sum")[0])

    def allgather_rows(self, a: np.ndarray, rows_per_rank) -> np.ndarray:
        """Concatenate every rank's rows (axis 0) in rank order; ranks may own different numbers of rows."""
        a = np.ascontiguousarray(a)
        if self.world == 1:
            return a.copy()
        most = max(rows_per_rank)
        pad = np.zeros((most,) + a.shape[1:], a.dtype)
        pad[: a.shape[0]] = a
        s = self.stage(pad.nbytes * (self.world + 1))
        s.upload(pad)
        recv = ctypes.c_void_p(s.ptr.value + pad.nbytes)
        self.allgather(s.ptr, recv, pad.nbytes, None)
        self.sync()
        allr = s.download((self.world + 1,) + pad.shape, a.dtype)[1:]
        return np.concatenate([allr[r][:n] for r, n in enumerate(rows_per_rank)], axis=0)

    def bcast_array(self, a, shape, dtype, root: int) -> np.ndarray:
        buf = np.ascontiguousarray(a, dtype=dtype).reshape(shape) if self.rank == root else np.zeros(shape, dtype)
        if self.world == 1:
            return buf
        s = self.stage(buf.nbytes)
        s.upload(buf)
        self.bcast(s.ptr, buf.nbytes, root, None)
        self.sync()
        return s.download(shape, dtype)

    def barrier(self):
        self.allreduce_sum(0)


class RcclComm(HostArrayOps):
    """ncclComm of `world` ranks; this process is `rank` and has its HIP device selected already."""

    def __init__(self, rank: int, world: int, comm_id: bytes):
        self.rank, self.world = int(rank), int(world)
        self._h = ctypes.c_void_p()
        idb = (ctypes.c_uint8 * 128).from_buffer_copy(comm_id)
        L.check(L.lib().ivx_comm_init(idb, self.rank, self.world, ctypes.byref(self._h)), "ivx_comm_init")
        self._stage = None

    @staticmethod
    def unique_id() -> bytes:
        idb = (ctypes.c_uint8 * 128)()
        L.check(L.lib().ivx_comm_unique_id(idb), "ivx_comm_unique_id")
        return bytes(idb)

    # -- pointer-level protocol -------------------------------------------------------------------------------
    def exchange(self, to_down, from_down, to_up, from_up, nbytes: int, stream):
        L.check(L.lib().ivx_comm_exchange(self._h, _vp(to_down), _vp(from_down), _vp(to_up), _vp(from_up),
                                          ctypes.c_size_t(int(nbytes)), stream), "ivx_comm_exchange")

    def exchange_vote(self, to_down, from_down, to_up, from_up, nbytes: int, vote, nvote: int, stream):
        L.check(L.lib().ivx_comm_exchange_vote(self._h, _vp(to_down), _vp(from_down), _vp(to_up), _vp(from_up),
                                               ctypes.c_size_t(int(nbytes)), _vp(vote), int(nvote), stream),
                "ivx_comm_exchange_vote")

    def allreduce(self, ptr, count: int, dtype: int, op: int, stream):
        L.check(L.lib().ivx_comm_allreduce(self._h, _vp(ptr), ctypes.c_size_t(int(count)), int(dtype), int(op), stream),
                "ivx_comm_allreduce")

    def allgather(self, send, recv, nbytes: int, stream):
        L.check(L.lib().ivx_comm_allgather(self._h, _vp(send), _vp(recv), ctypes.c_size_t(int(nbytes)), stream),
                "ivx_comm_allgather")

    def selftest(self, stream=None):
        """every ivx_comm_* entry point once on small buffers against the analytic answer (collective: all ranks call
        it); raises RuntimeError naming the collective that failed"""
        L.check(L.lib().ivx_comm_selftest(self._h, stream), "ivx_comm_selftest")

    def bcast(self, ptr, nbytes: int, root: int, stream):
        L.check(L.lib().ivx_comm_bcast(self._h, _vp(ptr), ctypes.c_size_t(int(nbytes)), int(root), stream), "ivx_comm_bcast")

    def send(self, ptr, nbytes: int, peer: int, stream):
        L.check(L.lib().ivx_comm_send(self._h, _vp(ptr), ctypes.c_size_t(int(nbytes)), int(peer), stream), "ivx_comm_send")

    def recv(self, ptr, nbytes: int, peer: int, stream):
        L.check(L.lib().ivx_comm_recv(self._h, _vp(ptr), ctypes.c_size_t(int(nbytes)), int(peer), stream), "ivx_comm_recv")

    def stage(self, nbytes: int):
        """A cached device buffer (grow-only) with upload / download, for the host-array conveniences."""
        from .device import DeviceBuffer

        if self._stage is None or self._stage.nbytes < nbytes:
            if self._stage is not None:
                self._stage.close()
            self._stage = DeviceBuffer(max(int(nbytes), 1 << 16))
        return self._stage

    def sync(self):
        L.synchronize()

    def close(self):
        if self._stage is not None:
            self._stage.close()
            self._stage = None
        if self._h:
            L.lib().ivx_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rendezvous_file() -> str:
    """Where rank 0 leaves the RCCL id: $IVX_COMM_FILE, else a name all ranks of ONE launch derive alike (the launcher's
    pid -- every rank is its child -- and MASTER_PORT)."""
    f = os.environ.get("IVX_COMM_FILE")
    if f:
        return f
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none").replace("/", "_")
    return os.path.join(os.environ.get("TMPDIR", "/tmp"),
                        "ivx_comm_%d_%s_%s.id" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), run))


def _nonce() -> bytes:
    """32 bytes every rank of ONE launch derives alike and a stale file of another launch cannot carry: the launcher's
    pid, the rendezvous port, the elastic run id and (when the launcher exports one, as bench.py does) IVX_COMM_NONCE."""
    import hashlib

    key = "%d|%s|%s|%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", ""),
                           os.environ.get("IVX_COMM_NONCE", ""))
    return hashlib.sha256(key.encode()).digest()


def exchange_id(rank: int, path: str, make_id, timeout_s: float = 300.0) -> bytes:
    """The rendezvous itself (no device needed, so the CPU suite runs it with the launcher's own environments): rank 0 makes
    the 128-byte id and leaves it at `path` behind this launch's nonce, every other rank polls for a file carrying that nonce."""
    t_start = time.time()
    if rank == 0:
        # a crashed run may have left an id behind under the same name: it must never be mistaken for this run's.  The
        # file is replaced atomically (written next to it, private, then renamed) and carries this launch's nonce in
        # front of the id.
        cid = make_id()
        tmp = "%s.%d.tmp" % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(_nonce() + cid)
        os.replace(tmp, path)
        return cid
    want = _nonce()
    # A launcher-made IVX_COMM_NONCE is unique per launch, so a matching file IS this launch's however late this rank
    # arrives (a slow first import on one rank must not time it out).  Without one (torchrun: launcher pid + port + run
    # id, which a recycled pid could repeat) the file must also be no older than this rank's own patience.
    oldest = -1.0 if os.environ.get("IVX_COMM_NONCE") else t_start - max(timeout_s, 120.0)
    while True:
        try:
            if os.path.getsize(path) == len(want) + 128 and os.path.getmtime(path) >= oldest:
                with open(path, "rb") as f:
                    blob = f.read()
                if blob[:len(want)] == want and len(blob) == len(want) + 128:
                    return blob[len(want):]
        except OSError:
            pass
        if time.time() - t_start > timeout_s:
            raise RuntimeError("rank %d: no RCCL id of this launch at %s after %.0f s" % (rank, path, timeout_s))
        time.sleep(0.02)


def init_from_env(timeout_s: float = 300.0) -> RcclComm:
    """RANK / WORLD_SIZE / LOCAL_RANK from the environment (torchrun's or bench.py's own launcher); selects the device,
    exchanges the id through `rendezvous_file()` and brings the communicator up."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    L.require_device()
    if L.device_count() <= local:
        raise RuntimeError("rank %d wants HIP device %d but only %d visible: one GPU per rank is required"
                           % (rank, local, L.device_count()))
    L.set_device(local)
    path = rendezvous_file()
    cid = exchange_id(rank, path, RcclComm.unique_id, timeout_s)
    comm = RcclComm(rank, world, cid)
    comm.barrier()
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return comm
