"""Z-slab sharding of the voxel path across GPUs: one process per GPU, RCCL (torch.distributed "nccl") between
Z-neighbours only.

The decomposition is the reference's own (SurfaceManager.AddNewActor, invesalius/data/surface.py:1362-1380: Z pieces
plus ONE overlap slice, concatenated):

* threshold            no communication (each rank thresholds its slab and its halo slices).
* marching cubes       rank g contours the cell layers between its slices and needs ONE slice of rank g+1 (its top
                       halo); pad_bottom only on rank 0, pad_top only on the last rank; soup = concatenation.
* region growing       local fix-point -> send the two interior boundary REACHED bit planes to the Z-neighbours ->
                       OR the received planes into the halo slices -> all-reduce(sum) of "words that gained bits"
                       -> repeat until 0.  Monotone, so it converges to exactly the single-GPU component.

The orchestration (`slab_region_grow`, `slab_layout`, `slab_mc_args`) is backend-agnostic: the GPU backend is
`SlabVolume` below; tests/test_parallel_gloo.py drives the same functions with a numpy backend over gloo
(world_size 2) and checks them against the single-volume oracle.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np


@dataclass
class SlabLayout:
    rank: int
    world: int
    nz: int          # interior slices owned by this rank
    hb: int          # 1 if a bottom halo slice (copy of rank-1's last slice) is stored
    ht: int          # 1 if a top halo slice (copy of rank+1's first slice) is stored
    z_global0: int   # global index of the first interior slice

    @property
    def local_dz(self) -> int:
        return self.nz + self.hb + self.ht

    @property
    def first_interior(self) -> int:
        return self.hb

    @property
    def last_interior(self) -> int:
        return self.hb + self.nz - 1


def slab_layout(rank: int, world: int, nz: int) -> SlabLayout:
    """Equal slabs of `nz` slices (weak scaling); halo slices towards existing neighbours only."""
    return SlabLayout(rank, world, nz, 1 if rank > 0 else 0, 1 if rank < world - 1 else 0, rank * nz)


def slab_mc_args(lay: SlabLayout, fill_border_holes: bool = True) -> dict:
    """Marching-cubes piece of this rank in the conventions of create_surface_piece (surface_process.py:96-103):
    roi = [z_global0, z_global0 + nz + 1) clipped to the volume, i.e. interior slices + the top halo slice."""
    return dict(z0=lay.hb, z1=lay.hb + lay.nz + lay.ht, roi_start=lay.z_global0,
                pad_bottom=(lay.rank == 0) and fill_border_holes,
                pad_top=(lay.rank == lay.world - 1) and fill_border_holes)


def local_seeds(lay: SlabLayout, seeds_xyz_global):
    """Global (x, y, z) seeds -> local coordinates of the seeds that fall inside this rank's stored slices."""
    out = []
    for x, y, z in seeds_xyz_global:
        zl = int(z) - lay.z_global0 + lay.hb
        if 0 <= zl < lay.local_dz:
            out.append((int(x), int(y), zl))
    return out


class DevPlane:
    """A block of device memory handed to / received from a communicator (anything with data_ptr() and nbytes)."""

    def __init__(self, ptr: int, nbytes: int, keep=None):
        self.ptr, self.nbytes, self._keep = int(ptr), int(nbytes), keep

    def data_ptr(self) -> int:
        return self.ptr


class TorchComm:
    """Neighbour exchange + scalar all-reduce over torch.distributed (RCCL on GPUs, gloo in the CPU tests).

    On GPUs, initialise torch's device (``torch.cuda.set_device`` / ``init_process_group``) BEFORE the first call into
    libivx: torch ships its own ROCm runtime and cannot find the GPU once the system runtime libivx links against has
    come up first in the process."""

    def __init__(self, dist, rank: int, world: int, device="cpu"):
        import torch

        self.dist, self.rank, self.world, self.device, self.torch = dist, rank, world, device, torch

    def _as_tensor(self, x):
        """DevPlane (raw HBM owned by libivx) -> a CUDA uint8 tensor RCCL can send; tensors pass through."""
        if x is None or not isinstance(x, DevPlane):
            return x
        from . import _lib as L

        t = self.torch.empty(x.nbytes, dtype=self.torch.uint8, device="cuda")
        L.check(L.lib().ivx_memcpy_d2d(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(x.ptr), ctypes.c_size_t(x.nbytes), None))
        L.synchronize()
        return t

    def exchange_host(self, to_down: np.ndarray, to_up: np.ndarray):
        """numpy planes (the one-time halo of the static input image)."""
        torch = self.torch
        mk = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)) if self.device != "cpu" else \
            (lambda a: torch.from_numpy(np.ascontiguousarray(a)))
        fd, fu = self.exchange(mk(to_down), mk(to_up))
        return (None if fd is None else fd.cpu().numpy()), (None if fu is None else fu.cpu().numpy())

    def exchange(self, to_down, to_up):
        """Send `to_down` to rank-1 and `to_up` to rank+1; returns (from_down, from_up) (None at the ends)."""
        torch, dist = self.torch, self.dist
        to_down, to_up = self._as_tensor(to_down), self._as_tensor(to_up)
        ops, from_down, from_up = [], None, None
        if self.rank > 0:
            from_down = torch.empty_like(to_down)
            ops += [dist.P2POp(dist.isend, to_down, self.rank - 1), dist.P2POp(dist.irecv, from_down, self.rank - 1)]
        if self.rank < self.world - 1:
            from_up = torch.empty_like(to_up)
            ops += [dist.P2POp(dist.isend, to_up, self.rank + 1), dist.P2POp(dist.irecv, from_up, self.rank + 1)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if self.device != "cpu":
                torch.cuda.current_stream().synchronize()
        return from_down, from_up

    def exchange_and_vote(self, to_down, to_up, changed: int):
        """The neighbour exchange and the vote in one collective: every rank contributes [changed | plane to the rank
        below | plane to the rank above] to an all-gather; returns (from_down, from_up, sum of everybody's `changed`).
        The planes are a few tens of KB, so gathering all of them costs nothing next to a second collective's latency."""
        torch, dist = self.torch, self.dist
        to_down, to_up = self._as_tensor(to_down), self._as_tensor(to_up)
        if self.world == 1:
            return None, None, int(changed)
        ref = to_down if to_down is not None else to_up
        if ref.dtype != torch.uint8:
            raise TypeError("exchange_and_vote: planes must be uint8 tensors")
        nb = int(ref.numel())
        rec = 16 + 2 * nb  # 16-byte header keeps the planes 16-byte aligned
        st = self.__dict__.setdefault("_gather", {})
        if st.get("nb") != nb:
            st["nb"] = nb
            st["send"] = torch.zeros(rec, dtype=torch.uint8, device=ref.device)
            st["recv"] = torch.empty(rec * self.world, dtype=torch.uint8, device=ref.device)
        send, recv = st["send"], st["recv"]
        send[:8].view(torch.int64).fill_(int(changed))  # a fill kernel, not a pageable host-to-device copy
        if to_down is not None:
            send[16:16 + nb] = to_down.reshape(-1)
        if to_up is not None:
            send[16 + nb:16 + 2 * nb] = to_up.reshape(-1)
        dist.all_gather_into_tensor(recv, send)
        recs = recv.view(self.world, rec)
        total = int(recs[:, :8].contiguous().view(torch.int64).sum().item())  # also waits for the gather to land
        from_down = recs[self.rank - 1, 16 + nb:16 + 2 * nb].reshape(ref.shape) if self.rank > 0 else None  # its "up" plane
        from_up = recs[self.rank + 1, 16:16 + nb].reshape(ref.shape) if self.rank < self.world - 1 else None  # its "down" plane
        return from_down, from_up, total

    def allreduce_sum(self, value: int) -> int:
        if self.world == 1:
            return int(value)
        if self.device == "cpu":
            t = self.torch.tensor([int(value)], dtype=self.torch.int64)
        else:  # one resident word, filled by a kernel: no pageable host-to-device copy per vote
            if getattr(self, "_vote", None) is None:
                self._vote = self.torch.zeros(1, dtype=self.torch.int64, device=self.device)
            t = self._vote
            t.fill_(int(value))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def plane_buffer(self, nbytes: int, slot: int):
        """A resident uint8 CUDA tensor the volume exports a boundary plane into directly (slot 0 = down, 1 = up): RCCL
        sends it as it is, no staging copy per exchange."""
        bufs = self.__dict__.setdefault("_planes", {})
        t = bufs.get(slot)
        if t is None or t.numel() != nbytes:
            t = bufs[slot] = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
        return t

    def allreduce_array(self, a: np.ndarray, op: str) -> np.ndarray:
        """Element-wise max / min / sum of one small host array per rank (a projection image)."""
        a = np.ascontiguousarray(a)
        wide = a.astype(np.int32) if a.dtype in (np.int16, np.uint16, np.uint8) else a  # neither gloo nor RCCL reduce 16-bit ints
        t = self.torch.from_numpy(wide)
        if self.device != "cpu":
            t = t.to(self.device)
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN,
                                    "sum": self.dist.ReduceOp.SUM}[op])
        return t.cpu().numpy().astype(a.dtype)

    def _bytes_tensor(self, a: np.ndarray):
        t = self.torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1))
        return t.to(self.device) if self.device != "cpu" else t

    def send_array(self, a: np.ndarray, to: int):
        """Blocking point-to-point send of a host array (a ray-state hand-over between Z-neighbours)."""
        self.dist.send(self._bytes_tensor(a), dst=to)

    def recv_array(self, shape, dtype, frm: int) -> np.ndarray:
        out = np.empty(shape, dtype)
        t = self._bytes_tensor(out)
        self.dist.recv(t, src=frm)
        return t.cpu().numpy().view(dtype).reshape(shape)

    def bcast_array(self, a: np.ndarray | None, shape, dtype, root: int) -> np.ndarray:
        buf = np.ascontiguousarray(a, dtype=dtype) if self.rank == root else np.empty(shape, dtype)
        t = self._bytes_tensor(buf)
        self.dist.broadcast(t, src=root)
        return t.cpu().numpy().view(dtype).reshape(shape)

    def allgather_rows(self, a: np.ndarray, rows_per_rank) -> np.ndarray:
        """Concatenate every rank's rows (axis 0) in rank order; ranks may own different numbers of rows."""
        torch = self.torch
        a = np.ascontiguousarray(a)
        most = max(rows_per_rank)
        pad = np.zeros((most,) + a.shape[1:], a.dtype)
        pad[: a.shape[0]] = a
        t = torch.from_numpy(pad.view(np.uint8).reshape(-1))  # bytes: every backend moves uint8
        if self.device != "cpu":
            t = t.to(self.device)
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return np.concatenate([p.cpu().numpy().view(a.dtype).reshape(pad.shape)[:n] for p, n in zip(parts, rows_per_rank)],
                              axis=0)


def slab_region_grow(backend, comm: TorchComm, lay: SlabLayout) -> int:
    """Iterate local fix-point + halo exchange to the global fix-point.  `backend` provides
    flood_run(), export_plane(z) -> tensor, or_plane(z, tensor) -> int (words/voxels that gained bits) and optionally
    or_planes(from_down, from_up) -> int.  Returns the number of exchange rounds.

    With a communicator that offers `exchange_and_vote` the "did anybody gain anything" vote of round k travels with the
    planes of round k+1 -- ONE collective per round instead of a neighbour exchange plus an all-reduce: the loop ends
    when a round reports that nobody gained anything in the round before (nobody flooded since, so the planes just
    exchanged are the ones everybody already had)."""
    rounds = 0
    gained = True  # the first pass floods from the seeds; later ones only where a neighbour's plane brought new bits
    merged = hasattr(comm, "exchange_and_vote")
    prev_changed = 1  # "something happened before round 1": the seeds
    down = up = None
    while True:
        if gained:
            backend.flood_run()
        if gained or rounds == 0:  # otherwise the planes exported last round are still current
            if hasattr(backend, "export_planes"):  # both planes, one stream wait
                down, up = backend.export_planes()
            else:
                down = backend.export_plane(lay.first_interior) if lay.hb else None
                up = backend.export_plane(lay.last_interior) if lay.ht else None
        if merged:
            from_down, from_up, total_prev = comm.exchange_and_vote(down, up, prev_changed)
            rounds += 1
            if total_prev == 0:
                return rounds
        else:
            from_down, from_up = comm.exchange(down, up)
        if hasattr(backend, "or_planes"):  # both planes, one read-back
            changed = backend.or_planes(from_down, from_up)
        else:
            changed = 0
            if from_down is not None:
                changed += backend.or_plane(0, from_down)
            if from_up is not None:
                changed += backend.or_plane(lay.local_dz - 1, from_up)
        gained = changed > 0
        if merged:
            prev_changed = changed
            continue
        rounds += 1
        if comm.allreduce_sum(changed) == 0:
            return rounds


def slab_project_combine(partial: np.ndarray, comm: TorchComm, axis: int, op: str, rows_per_rank, global_dz: int,
                         gather: bool = True) -> np.ndarray:
    """MaxIP / MinIP / MeanIP of a Z-sharded volume (`slice_.py:885-889,969-973,1056-1060`).
    `partial` is this rank's reduction over its OWN slices (halo slices excluded):
      axis 0 (rays along Z, every rank holds a piece of every ray): op "max" / "min" -> the image dtype, all-reduced
             with max / min; op "mean" -> exact int64 sums, all-reduced with sum, then divided by the global depth in
             float64 (sums of < 2^53 are exact, so this is numpy's mean to the last bit);
      axis 1 / 2 (rays inside a slice): the rank's rows are final; gather=True concatenates all ranks' rows."""
    if axis == 0:
        if op == "mean":
            return comm.allreduce_array(partial.astype(np.int64), "sum").astype(np.float64) / float(global_dz)
        return comm.allreduce_array(partial, op)
    return comm.allgather_rows(partial, rows_per_rank) if gather else partial


# ---------------------------------------------------------------------------------------------------------------
# GPU backend
# ---------------------------------------------------------------------------------------------------------------
def _make_slab_volume():
    from . import _lib as L
    from .device import DeviceBuffer, DeviceVolume, c64

    class SlabVolume(DeviceVolume):
        """This rank's Z-slab (+ halo slices) resident in HBM.  Same call surface as DeviceVolume; region growing and
        marching cubes are the sharded versions."""

        def __init__(self, image_slab: np.ndarray, rank: int, world: int, dist=None, spacing=(1.0, 1.0, 1.0), comm=None,
                     device=None):
            self.lay = slab_layout(rank, world, image_slab.shape[0])
            # `comm` may be injected (tests/test_gpu_slab.py drives several ranks on ONE GPU through an in-process
            # loop-back that has the same exchange / allreduce_sum interface as TorchComm)
            self.comm = comm if comm is not None else TorchComm(dist, rank, world, device="cuda")
            # one-time halo exchange of the IMAGE (static input): my first slice goes down, my last slice goes up
            from_down, from_up = self.comm.exchange_host(image_slab[0], image_slab[-1])
            parts = []
            if self.lay.hb:
                parts.append(np.asarray(from_down).reshape(image_slab.shape[1:])[None])
            parts.append(image_slab)
            if self.lay.ht:
                parts.append(np.asarray(from_up).reshape(image_slab.shape[1:])[None])
            local = np.concatenate(parts) if len(parts) > 1 else image_slab
            super().__init__(np.ascontiguousarray(local), spacing=spacing, device=device)
            self.plane_words = self.dy * self.plan.wx
            self._send = [DeviceBuffer(self.plane_words * 8) for _ in range(2)]
            self._cand = self.cand

        # -- backend protocol of slab_region_grow ---------------------------------------------------------------
        def flood_run(self):
            r = ctypes.c_int(0)
            L.check(L.lib().ivx_dev_flood_run(ctypes.byref(self.plan), self._cand.ptr, self.reached.ptr,
                                              self.flood_scratch.ptr, ctypes.byref(r), self.stream), "flood_run")
            self._rounds += r.value

        def export_plane(self, z: int):
            slot = 0 if z == self.lay.first_interior else 1
            nb = self.plane_words * 8
            if hasattr(self.comm, "plane_buffer") and self.comm.device != "cpu":
                # straight into the tensor RCCL sends (TorchComm): one device copy, one stream wait
                t = self.comm.plane_buffer(nb, slot)
                L.check(L.lib().ivx_memcpy_d2d(ctypes.c_void_p(t.data_ptr()), self.reached.at(z * nb), ctypes.c_size_t(nb),
                                               self.stream))
                self.sync()  # complete before the communicator (RCCL runs on torch's stream) reads it
                return t
            b = self._send[slot]
            L.check(L.lib().ivx_memcpy_d2d(b.ptr, self.reached.at(z * nb), ctypes.c_size_t(nb), self.stream))
            self.sync()
            return DevPlane(b.ptr.value, nb, keep=b)

        def export_planes(self):
            """(plane for the rank below, plane for the rank above), None at the ends; ONE stream wait for both"""
            nb = self.plane_words * 8
            direct = hasattr(self.comm, "plane_buffer") and self.comm.device != "cpu"
            out = []
            for slot, (have, z) in enumerate(((self.lay.hb, self.lay.first_interior), (self.lay.ht, self.lay.last_interior))):
                if not have:
                    out.append(None)
                    continue
                if direct:
                    t = self.comm.plane_buffer(nb, slot)
                    dst = ctypes.c_void_p(t.data_ptr())
                    out.append(t)
                else:
                    b = self._send[slot]
                    dst = b.ptr
                    out.append(DevPlane(b.ptr.value, nb, keep=b))
                L.check(L.lib().ivx_memcpy_d2d(dst, self.reached.at(z * nb), ctypes.c_size_t(nb), self.stream))
            if out[0] is not None or out[1] is not None:
                self.sync()  # complete before the communicator (RCCL runs on torch's stream) reads them
            return out[0], out[1]

        def or_planes(self, from_down, from_up) -> int:
            chg = ctypes.c_int(0)
            pd = ctypes.c_void_p(from_down.data_ptr()) if from_down is not None else None
            pu = ctypes.c_void_p(from_up.data_ptr()) if from_up is not None else None
            L.check(L.lib().ivx_dev_flood_or_planes(ctypes.byref(self.plan), self._cand.ptr, self.reached.ptr, c64(0), pd,
                                                    c64(self.lay.local_dz - 1), pu, self.flood_scratch.ptr,
                                                    ctypes.byref(chg), self.stream), "flood_or_planes")
            return chg.value

        def or_plane(self, z: int, tensor) -> int:
            chg = ctypes.c_int(0)
            L.check(L.lib().ivx_dev_flood_or_plane(ctypes.byref(self.plan), self._cand.ptr, self.reached.ptr, c64(z),
                                                   ctypes.c_void_p(tensor.data_ptr()), self.flood_scratch.ptr,
                                                   ctypes.byref(chg), self.stream), "flood_or_plane")
            return chg.value

        # -- sharded operations ----------------------------------------------------------------------------------
        def region_grow(self, seeds_xyz_global, t0, t1, strct, fill: int = 1, select_value=254) -> int:
            """Seeds are GLOBAL (x, y, z); z counts slices of the whole (world * nz)-slice volume."""
            lib = L.lib()
            s3 = np.ascontiguousarray(strct, dtype=np.uint8)
            bits = ctypes.c_uint32(0)
            L.check(lib.ivx_flood_strct_bits(L.ptr(s3), L.i64(s3.shape), ctypes.byref(bits)))
            self.plan.strct_bits = bits.value
            p, st = ctypes.byref(self.plan), self.stream
            t0, t1 = float(int(t0)), float(int(t1))
            self._before_flood()
            L.check(lib.ivx_dev_flood_clear(p, self.reached.ptr, self.flood_scratch.ptr, st))
            # same shortcut as the single-GPU pipeline: the plane the threshold pass left behind, when it provably IS
            # the candidate plane of this slab (halo slices included: they were thresholded with the slab)
            self._cand, shared = self._candidate_plane(None, t0, t1, fill)
            loc = local_seeds(self.lay, seeds_xyz_global)
            if loc:
                seeds = np.ascontiguousarray(np.array(loc, dtype=np.int64).reshape(-1, 3))
                L.check(lib.ivx_dev_flood_seed(p, L.I16, self.image.raw, ctypes.c_double(t0), ctypes.c_double(t1),
                                               L.ptr(seeds), c64(len(seeds)), self._cand.ptr, self.reached.ptr,
                                               self.flood_scratch.ptr, st), "region_grow")
            self._rounds = 0
            slab_region_grow(self, self.comm, self.lay)
            self._gate_armed = False  # an armed gate has been opened by the first local flood
            self._apply_reached(fill, select_value, shared)
            return self._rounds

        def reached_count(self) -> int:
            """Reached voxels of the INTERIOR slices only (halo slices belong to the neighbours)."""
            sub = L.FloodPlan(self.lay.nz, self.dy, self.dx, self.plan.wx, self.plan.strct_bits)
            n = ctypes.c_int64(0)
            L.check(L.lib().ivx_dev_flood_count(ctypes.byref(sub), self.reached.at(self.lay.hb * self.plane_words * 8),
                                                ctypes.byref(n), self.stream))
            return n.value

        def project_global(self, axis: int, op: str, gather: bool = True, source=None) -> np.ndarray:
            """MaxIP ("max") / MinIP ("min") / MeanIP ("mean") of the WHOLE volume along `axis`, from every rank's
            resident slab: local reduce over the interior slices on the GPU, then slab_project_combine."""
            lay = self.lay
            nint = lay.last_interior - lay.first_interior + 1
            code = {"max": L.MIP_MAX, "min": L.MIP_MIN, "mean": L.MIP_SUM if axis == 0 else L.MIP_MEAN}[op]
            oshape = (self.dy, self.dx) if axis == 0 else ((nint, self.dx) if axis == 1 else (nint, self.dy))
            odt = np.int64 if code == L.MIP_SUM else (np.float64 if code == L.MIP_MEAN else np.int16)
            out = DeviceBuffer(int(np.prod(oshape)) * np.dtype(odt).itemsize + 16)
            off = lay.first_interior * self.dy * self.dx * 2
            src = self.image.raw_at(off) if source is None else source.at(off)  # `source`: another resident int16 slab
            L.check(L.lib().ivx_dev_mip_reduce(L.I16, src, c64(nint), c64(self.dy), c64(self.dx), int(axis), int(code), out.ptr,
                                               self.stream), "project")
            self.sync()
            partial = out.download(oshape, odt)
            out.close()
            if lay.world == 1:
                return partial.astype(np.float64) / float(nint) if code == L.MIP_SUM else partial
            rows = [lay.nz] * lay.world
            return slab_project_combine(partial, self.comm, axis, op, rows, lay.nz * lay.world, gather)

        def rays_global(self, kind: str, axis: int, p0, p1, gather: bool = True, source=None) -> np.ndarray:
            """LMIP ("lmip": p0, p1 = tmin, tmax) or MIDA ("mida": p0, p1 = wl, ww) of the WHOLE volume along `axis`
            (invesalius_rs lmip / mida, mips.rs:7-168).  Rays inside a slice (axis 1, 2) are rank-local rows, gathered.
            Rays along Z are order-dependent: the slabs are walked front to back, each rank resumes every ray from the
            state its lower neighbour hands over (5 doubles per pixel) and the last rank's image is broadcast.  MIDA
            normalises by the min / max of the whole volume: one all-reduce of two numbers first."""
            lay = self.lay
            code = {"lmip": 0, "mida": 1}[kind]
            nint = lay.nz
            off = lay.first_interior * self.dy * self.dx * 2
            src = self.image.raw_at(off) if source is None else source.at(off)
            lib = L.lib()
            mm = DeviceBuffer(64)
            status = DeviceBuffer(64)
            status.zero(self.stream)
            if code == 1:
                L.check(lib.ivx_dev_minmax_f32(L.I16, src, c64(nint * self.dy * self.dx), mm.ptr, self.stream))
                self.sync()
                local = mm.download((2,), np.float32).astype(np.float64)
                if lay.world > 1:
                    lo = self.comm.allreduce_array(local[:1], "min")
                    hi = self.comm.allreduce_array(local[1:], "max")
                    mm.upload(np.array([lo[0], hi[0]], np.float32))
            p0, p1 = float(int(p0)), float(int(p1))  # the wrappers' int() truncation (invesalius_rs/__init__.py:91-95)
            if axis != 0:
                oshape = (nint, self.dx) if axis == 1 else (nint, self.dy)
                out = DeviceBuffer(int(np.prod(oshape)) * 2 + 16)
                if code == 0:
                    L.check(lib.ivx_dev_lmip(L.I16, src, c64(nint), c64(self.dy), c64(self.dx), int(axis), ctypes.c_double(p0),
                                             ctypes.c_double(p1), out.ptr, self.stream), "lmip")
                else:
                    L.check(lib.ivx_dev_mida(L.I16, src, c64(nint), c64(self.dy), c64(self.dx), int(axis), ctypes.c_float(p0),
                                             ctypes.c_float(p1), mm.ptr, L.I16, out.ptr, status.ptr, self.stream), "mida")
                self.sync()
                rows = out.download(oshape, np.int16)
                bad = status.download((1,), np.int32)[0]
                for b in (out, mm, status):
                    b.close()
                if bad:
                    raise ValueError("mida: a result does not fit the output dtype (the reference's NumCast panics)")
                if lay.world == 1 or not gather:
                    return rows
                return self.comm.allgather_rows(rows, [lay.nz] * lay.world)
            npix = self.dy * self.dx
            state = DeviceBuffer(npix * 5 * 8 + 16)
            out = DeviceBuffer(npix * 2 + 16)
            first, last = lay.rank == 0, lay.rank == lay.world - 1
            if not first:
                state.upload(self.comm.recv_array((npix * 5,), np.float64, lay.rank - 1))
            L.check(lib.ivx_dev_rays_z_slab(code, L.I16, src, c64(nint), c64(self.dy), c64(self.dx), ctypes.c_double(p0),
                                            ctypes.c_double(p1), mm.ptr, None if first else state.ptr, None if last else state.ptr,
                                            L.I16, out.ptr, status.ptr, self.stream), "rays_z_slab")
            self.sync()
            if not last:
                self.comm.send_array(state.download((npix * 5,), np.float64), lay.rank + 1)
            img = out.download((self.dy, self.dx), np.int16) if last else None
            bad = int(status.download((1,), np.int32)[0]) if last else 0
            for b in (state, out, mm, status):
                b.close()
            if lay.world > 1:
                flag = self.comm.bcast_array(np.array([bad], np.int32) if last else None, (1,), np.int32, lay.world - 1)
                bad = int(flag[0])
                img = self.comm.bcast_array(img, (self.dy, self.dx), np.int16, lay.world - 1)
            if bad:
                raise ValueError("mida: a result does not fit the output dtype (the reference's NumCast panics)")
            return img

        def fast_countour_mip_global(self, n: float, axis: int, wl, ww, tmip: int) -> np.ndarray:
            """fast_countour_mip (mips.rs:215-279) of the whole volume: the contour volume is computed on the stored
            slices -- the halo slices give every owned voxel its true z neighbours, the clamped differences at the ends
            of the volume are the unsharded ones -- and then projected like any other volume (tmip 0 = MaxIP,
            1 = LMIP(700, 3033), 2 = MIDA)."""
            tmp, status = DeviceBuffer(self.n * 2 + 16), DeviceBuffer(64)
            status.zero(self.stream)
            L.check(L.lib().ivx_dev_fcm_volume(L.I16, self.image.raw, c64(self.dz), c64(self.dy), c64(self.dx), ctypes.c_float(n),
                                               int(axis), tmp.ptr, status.ptr, self.stream), "fcm_volume")
            self.sync()
            bad = int(status.download((1,), np.int32)[0])
            if self.lay.world > 1:
                bad = int(self.comm.allreduce_array(np.array([bad], np.int64), "max")[0])
            if bad:
                tmp.close()
                status.close()
                raise ValueError("fast_countour_mip: a contour value does not fit the image dtype (the reference's NumCast panics)")
            try:
                if tmip == 0:
                    return self.project_global(axis, "max", source=tmp)
                if tmip == 1:
                    return self.rays_global("lmip", axis, 700, 3033, source=tmp)
                if tmip == 2:
                    return self.rays_global("mida", axis, wl, ww, source=tmp)
                raise ValueError("tmip must be 0, 1 or 2")
            finally:
                tmp.close()
                status.close()

        def _surface_params(self, from_binary, min_value, max_value, fill_border_holes):
            """this rank's piece: its cell layers plus the one slice of the upper neighbour (the reference's o_piece = 1)"""
            a = slab_mc_args(self.lay, fill_border_holes)
            p = self._mc_params(from_binary, min_value, max_value, fill_border_holes, z0=a["z0"], z1=a["z1"],
                                roi_start=a["roi_start"], pad_bottom=a["pad_bottom"], pad_top=a["pad_top"])
            return p, a["z0"]

        def marching_cubes(self, from_binary=True, min_value=0, max_value=0, fill_border_holes=True, download=False):
            return super().marching_cubes(from_binary, min_value, max_value, fill_border_holes, download)

        def marching_cubes_indexed(self, from_binary=True, min_value=0, max_value=0, fill_border_holes=True, download=False):
            """This rank's piece as an indexed mesh.  Vertices on the plane shared with the next rank exist on both
            sides; `stitch_piece_meshes` merges them (the vtkCleanPolyData step of join_process_surface)."""
            return super().marching_cubes_indexed(from_binary, min_value, max_value, fill_border_holes, download)

    return SlabVolume


def stitch_piece_meshes(pieces):
    """Cross-slab stitch (vtkAppendPolyData + vtkCleanPolyData in surface_process.py:229-268): concatenate the
    ranks' indexed pieces ``[(verts (V,3) float32, faces (T,3) int32), ...]`` in rank order and merge the vertices two
    consecutive pieces both carry on their shared plane.  Both sides computed those vertices with the same arithmetic
    on the same voxels, so they are equal bit for bit and the merge is an exact match on the float32 triple -- looked
    for only among the vertices of the two pieces that sit on the shared plane's z (a few thousand per plane).
    Returns (verts, faces) with the triangles in rank order."""
    out_v, out_f = [], []
    base = 0
    prev = None  # (global ids, verts) of the previous piece, for the plane look-up
    for verts, faces in pieces:
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
        gid = np.arange(len(verts), dtype=np.int64) + base
        keep = np.ones(len(verts), bool)
        if prev is not None and len(verts) and len(prev[1]):
            pg, pv = prev
            z_shared = np.intersect1d(np.unique(pv[:, 2]), np.unique(verts[:, 2]))
            if len(z_shared):
                a = np.isin(pv[:, 2], z_shared)
                b = np.isin(verts[:, 2], z_shared)
                key = lambda v: np.ascontiguousarray(v).view([("", np.uint32)] * 3).ravel()
                ka, kb = key(pv[a].view(np.uint32)), key(verts[b].view(np.uint32))
                order = np.argsort(ka)
                pos = np.searchsorted(ka[order], kb)
                pos[pos >= len(ka)] = 0
                hit = (len(ka) > 0) & (ka[order][pos] == kb) if len(ka) else np.zeros(len(kb), bool)
                idx_b = np.nonzero(b)[0][hit]
                gid[idx_b] = pg[a][order][pos[hit]]
                keep[idx_b] = False
        # compact: the surviving vertices of this piece get consecutive global ids after `base`
        new_ids = np.cumsum(keep) - 1 + base
        gid = np.where(keep, new_ids, gid)
        out_v.append(verts[keep])
        out_f.append(gid[faces] if len(faces) else faces)
        base += int(keep.sum())
        prev = (gid, verts)
    if not out_v:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    return np.concatenate(out_v), np.concatenate(out_f).astype(np.int32)


def __getattr__(name):  # SlabVolume needs libivx + a device; build the class lazily so the CPU tests can import us
    if name == "SlabVolume":
        cls = _make_slab_volume()
        globals()["SlabVolume"] = cls
        return cls
    raise AttributeError(name)
