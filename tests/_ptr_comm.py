"""Test stand-ins for invesalius3_amd.comm.RcclComm (test infrastructure): the same pointer-level protocol
(exchange / exchange_vote / allreduce / allgather / bcast / send / recv / stage / sync) over

* GlooPtrComm   host pointers + torch.distributed gloo: world_size-N CPU tests of the slab orchestration;
* LoopbackWorld device (or host) pointers, N ranks as threads of ONE process on ONE GPU: the real HIP slab backend
                without RCCL (RCCL refuses two ranks on one device).
Only tests import this; the product's communicator is RCCL behind the C ABI."""
import ctypes
import threading

import numpy as np

from invesalius3_amd.comm import HostArrayOps

_NP = {5: np.int32, 6: np.int64, 4: np.float32, 2: np.float64, 0: np.uint8, 7: np.int8}  # IVX dtype codes


def _addr(p):
    if p is None:
        return 0
    return int(p.value or 0) if isinstance(p, ctypes.c_void_p) else int(p)


def host_view(p, nbytes, dtype=np.uint8):
    a = _addr(p)
    return np.ctypeslib.as_array((ctypes.c_uint8 * int(nbytes)).from_address(a)).view(dtype)


class HostStage:
    def __init__(self, nbytes):
        self.buf = np.zeros(int(nbytes), np.uint8)
        self.nbytes = int(nbytes)
        self.ptr = ctypes.c_void_p(self.buf.ctypes.data)

    def upload(self, a):
        a = np.ascontiguousarray(a)
        self.buf[: a.nbytes] = a.view(np.uint8).reshape(-1)

    def download(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.buf[:n].view(dtype).reshape(shape).copy()


class GlooPtrComm(HostArrayOps):
    """pointers are HOST addresses; gloo moves the bytes"""

    def __init__(self, dist, rank, world):
        import torch
        self.dist, self.rank, self.world, self.torch = dist, rank, world, torch
        self._stage = None

    def _t(self, p, nbytes, dtype=np.uint8):
        return self.torch.from_numpy(host_view(p, nbytes, dtype))

    def exchange(self, to_down, from_down, to_up, from_up, nbytes, stream):
        self.exchange_vote(to_down, from_down, to_up, from_up, nbytes, None, 0, stream)

    def exchange_vote(self, to_down, from_down, to_up, from_up, nbytes, vote, nvote, stream):
        dist = self.dist
        ops = []
        if self.world > 1 and nbytes:
            if self.rank > 0:
                if _addr(to_down):
                    ops.append(dist.P2POp(dist.isend, self._t(to_down, nbytes), self.rank - 1))
                if _addr(from_down):
                    ops.append(dist.P2POp(dist.irecv, self._t(from_down, nbytes), self.rank - 1))
            if self.rank < self.world - 1:
                if _addr(to_up):
                    ops.append(dist.P2POp(dist.isend, self._t(to_up, nbytes), self.rank + 1))
                if _addr(from_up):
                    ops.append(dist.P2POp(dist.irecv, self._t(from_up, nbytes), self.rank + 1))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.world > 1 and _addr(vote) and nvote:
            dist.all_reduce(self._t(vote, 4 * nvote, np.int32), op=dist.ReduceOp.SUM)

    def allreduce(self, ptr, count, dtype, op, stream):
        if self.world == 1:
            return
        dt = _NP[dtype]
        t = self._t(ptr, count * np.dtype(dt).itemsize, dt)
        self.dist.all_reduce(t, op=[self.dist.ReduceOp.SUM, self.dist.ReduceOp.MAX, self.dist.ReduceOp.MIN][op])

    def allgather(self, send, recv, nbytes, stream):
        parts = [self._t(_addr(recv) + r * nbytes, nbytes) for r in range(self.world)]
        self.dist.all_gather(parts, self._t(send, nbytes).clone())

    def bcast(self, ptr, nbytes, root, stream):
        self.dist.broadcast(self._t(ptr, nbytes), src=root)

    def send(self, ptr, nbytes, peer, stream):
        self.dist.send(self._t(ptr, nbytes), dst=peer)

    def recv(self, ptr, nbytes, peer, stream):
        self.dist.recv(self._t(ptr, nbytes), src=peer)

    def stage(self, nbytes):
        if self._stage is None or self._stage.nbytes < nbytes:
            self._stage = HostStage(max(nbytes, 4096))
        return self._stage

    def sync(self):
        pass


class LoopbackWorld:
    """N ranks = N threads of this process.  `device=True`: pointers are HIP device addresses of ONE GPU (copies through
    ivx_memcpy_d2d / d2h / h2d after a device synchronise); `device=False`: host addresses."""

    def __init__(self, world, device=True):
        self.world, self.device = world, device
        self.barrier = threading.Barrier(world)
        self.box = {}
        self.cv = threading.Condition()
        self.p2p = {}
        self.collectives = 0

    def comm(self, rank):
        return LoopbackComm(self, rank)


class LoopbackComm(HostArrayOps):
    def __init__(self, w, rank):
        self.w, self.rank, self.world = w, rank, w.world
        self._stage = None

    # raw byte movers ----------------------------------------------------------------------------------------
    def _read(self, p, nbytes):
        if not self.w.device:
            return host_view(p, nbytes).copy()
        from invesalius3_amd import _lib as L
        out = np.empty(int(nbytes), np.uint8)
        L.check(L.lib().ivx_memcpy_d2h(L.ptr(out), ctypes.c_void_p(_addr(p)), ctypes.c_size_t(int(nbytes))))
        return out

    def _write(self, p, data):
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        if not self.w.device:
            host_view(p, data.nbytes)[:] = data
            return
        from invesalius3_amd import _lib as L
        L.check(L.lib().ivx_memcpy_h2d(ctypes.c_void_p(_addr(p)), L.ptr(data), ctypes.c_size_t(data.nbytes)))

    def sync(self):
        if self.w.device:
            from invesalius3_amd import _lib as L
            L.synchronize()

    # protocol -----------------------------------------------------------------------------------------------
    def exchange(self, to_down, from_down, to_up, from_up, nbytes, stream):
        self.exchange_vote(to_down, from_down, to_up, from_up, nbytes, None, 0, stream)

    def exchange_vote(self, to_down, from_down, to_up, from_up, nbytes, vote, nvote, stream):
        w = self.w
        self.sync()  # everything this rank queued has landed before a peer reads it
        if self.rank == 0:
            w.collectives += 1
        if nbytes:
            if self.rank > 0 and _addr(to_down):
                w.box[(self.rank, "down")] = self._read(to_down, nbytes)
            if self.rank < self.world - 1 and _addr(to_up):
                w.box[(self.rank, "up")] = self._read(to_up, nbytes)
        if _addr(vote) and nvote:
            w.box[(self.rank, "vote")] = self._read(vote, 4 * nvote).view(np.int32)
        w.barrier.wait()
        if nbytes:
            if self.rank > 0 and _addr(from_down):
                self._write(from_down, w.box[(self.rank - 1, "up")])
            if self.rank < self.world - 1 and _addr(from_up):
                self._write(from_up, w.box[(self.rank + 1, "down")])
        if _addr(vote) and nvote:
            self._write(vote, np.add.reduce([w.box[(r, "vote")] for r in range(self.world)]).astype(np.int32))
        w.barrier.wait()

    def allreduce(self, ptr, count, dtype, op, stream):
        w = self.w
        dt = _NP[dtype]
        self.sync()
        w.box[(self.rank, "ar")] = self._read(ptr, count * np.dtype(dt).itemsize).view(dt)
        w.barrier.wait()
        parts = [w.box[(r, "ar")] for r in range(self.world)]
        out = [np.add.reduce, np.maximum.reduce, np.minimum.reduce][op](parts).astype(dt)
        w.barrier.wait()
        self._write(ptr, out)

    def allgather(self, send, recv, nbytes, stream):
        w = self.w
        self.sync()
        w.box[(self.rank, "ag")] = self._read(send, nbytes)
        w.barrier.wait()
        out = np.concatenate([w.box[(r, "ag")] for r in range(self.world)])
        w.barrier.wait()
        self._write(recv, out)

    def bcast(self, ptr, nbytes, root, stream):
        w = self.w
        self.sync()
        if self.rank == root:
            w.box[("bc", root)] = self._read(ptr, nbytes)
        w.barrier.wait()
        out = w.box[("bc", root)]
        w.barrier.wait()
        if self.rank != root:
            self._write(ptr, out)

    def send(self, ptr, nbytes, peer, stream):
        self.sync()
        data = self._read(ptr, nbytes)
        with self.w.cv:
            self.w.p2p.setdefault((self.rank, peer), []).append(data)
            self.w.cv.notify_all()

    def recv(self, ptr, nbytes, peer, stream):
        w = self.w
        with w.cv:
            assert w.cv.wait_for(lambda: w.p2p.get((peer, self.rank)), timeout=120), "loop-back recv timed out"
            data = w.p2p[(peer, self.rank)].pop(0)
        self.sync()
        self._write(ptr, data)

    def stage(self, nbytes):
        if self.w.device:
            from invesalius3_amd.device import DeviceBuffer
            if self._stage is None or self._stage.nbytes < nbytes:
                self._stage = DeviceBuffer(max(int(nbytes), 1 << 16))
            return self._stage
        if self._stage is None or self._stage.nbytes < nbytes:
            self._stage = HostStage(max(nbytes, 4096))
        return self._stage
