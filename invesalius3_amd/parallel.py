"""Z-slab sharding of the voxel path across GPUs: one process per GPU, RCCL between Z-neighbours through the C ABI
(`invesalius3_amd.comm.RcclComm` -> ivx_comm_*, csrc/ivx_comm.hip); no PyTorch.

The decomposition is the reference's own (SurfaceManager.AddNewActor, invesalius/data/surface.py:1362-1380: Z pieces
plus ONE overlap slice, concatenated):

* threshold            no communication (each rank thresholds its slab and its halo slices).
* marching cubes       rank g contours the cell layers between its slices and needs ONE slice of rank g+1 (its top
                       halo); pad_bottom only on rank 0, pad_top only on the last rank; soup = concatenation.
* region growing       local fix-point -> ncclSend/Recv of the two interior boundary REACHED bit planes to the
                       Z-neighbours + a 4-byte all-reduce of "words that gained bits", one enqueue-only call on the kernels' stream
                       -> OR the received planes into the halo slices -> one host read -> repeat until nobody gained.
                       Monotone, so it converges to exactly the single-GPU component.
* projections          rays inside a slice are rank-local rows (all-gather); rays along Z: MaxIP / MinIP / MeanIP
                       all-reduce one image, LMIP / MIDA hand the per-ray state from slab to slab device-to-device.

The orchestration (`slab_region_grow`, `slab_layout`, `slab_mc_args`, `slab_project_combine`) is written against the
communicator's pointer-level protocol (comm.py): the GPU backend is `SlabVolume` below over `RcclComm`;
tests/test_parallel_gloo.py drives the same functions with a numpy backend over gloo (world_size 2 and 3, host
pointers) and tests/test_gpu_slab.py with several ranks on ONE GPU through an in-process loop-back.
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


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOSPAN = _NoSpan()


def slab_region_grow(backend, comm, lay: SlabLayout) -> int:
    """Iterate local fix-point + halo exchange to the global fix-point; returns the number of exchange rounds.

    One collective and ONE host read per round.  `backend` provides
      flood_run()            local fix-point from whatever is reached / was OR-ed in;
      round_ptrs()        -> (to_down, from_down, to_up, from_up, nbytes, vote, stream): the two interior boundary planes
                             of the reached set (sent where they lie), two receive planes, and an int32 word;
      stage_vote()           vote <- "words that gained bits in my last OR" (1 before the first round: the seeds);
      or_planes()            OR the received planes into the halo slices, count the words that gained bits;
      read_votes()        -> (everybody's staged votes summed, my new count): the round's only host read.
    `comm.exchange_vote` moves the planes between Z-neighbours and all-reduces the vote word right behind them, so the
    "did anybody gain anything" vote of round k travels with the planes of round k+1: the loop ends when a round
    reports that nobody gained anything in the round before (nobody flooded since, so the planes just exchanged are
    the ones everybody already had).  Monotone, so it converges to exactly the single-volume component."""
    rounds = 0
    gained = True  # the first pass floods from the seeds; later ones only where a neighbour's plane brought new bits
    while True:
        if gained:
            backend.flood_run()
        backend.stage_vote()
        to_down, from_down, to_up, from_up, nbytes, vote, stream = backend.round_ptrs()
        # (HIP events around the collective when the backend brackets stages: `comm_exchange_vote` of the per-stage table is
        # the device time of the plane exchange + vote all-reduce, summed over the rounds of one flood)
        with (backend.comm_span("comm_exchange_vote") if hasattr(backend, "comm_span") else _NOSPAN):
            comm.exchange_vote(to_down if lay.hb else None, from_down if lay.hb else None, to_up if lay.ht else None,
                               from_up if lay.ht else None, nbytes, vote, 1, stream)
        backend.or_planes()
        rounds += 1
        total_prev, changed = backend.read_votes()
        if total_prev == 0:
            return rounds
        gained = changed > 0


def slab_project_combine(partial: np.ndarray, comm, axis: int, op: str, rows_per_rank, global_dz: int,
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
    from .comm import HostArrayOps

    class _SoloComm(HostArrayOps):
        """world of one: nothing to exchange"""
        rank, world = 0, 1

        def exchange(self, *a):
            pass

        def exchange_vote(self, *a):
            pass

        def sync(self):
            pass

    class SlabVolume(DeviceVolume):
        """This rank's Z-slab (+ halo slices) resident in HBM.  Same call surface as DeviceVolume; region growing and
        marching cubes are the sharded versions."""

        def __init__(self, image_slab, rank: int, world: int, comm=None, spacing=(1.0, 1.0, 1.0), device=None, shape=None,
                     fill=None):
            """`image_slab`: this rank's (nz, dy, dx) int16 slices on the host -- or None with `shape` = that shape and
            `fill(device_pointer, stream)` a callable that makes the slices in HBM where they are used (bench.py's
            configs[3]: every rank synthesises its own 2 GB slab)."""
            if image_slab is None:
                if shape is None or fill is None:
                    raise ValueError("SlabVolume: pass an image slab, or shape= and fill=")
                slab_shape = tuple(int(v) for v in shape)
            else:
                if image_slab.dtype != np.int16 or image_slab.ndim != 3:
                    raise TypeError("image slab must be a 3-D int16 array")
                slab_shape = tuple(image_slab.shape)
            self.lay = lay = slab_layout(rank, world, slab_shape[0])
            if comm is None:
                if world != 1:
                    raise ValueError("SlabVolume: a communicator is required for world > 1 (invesalius3_amd.comm.init_from_env)")
                comm = _SoloComm()
            self.comm = comm
            super().__init__(None, shape=(lay.local_dz,) + slab_shape[1:], spacing=spacing, device=device)
            # my slices go to their place in HBM; the halo slices of the IMAGE (static input) come from the Z-neighbours
            # device to device: my first slice goes down, my last slice goes up (the reference's o_piece = 1)
            sb = self.dy * self.dx * 2
            if image_slab is None:
                fill(self.image.raw_at(lay.hb * sb), self.stream)
            else:
                img = np.ascontiguousarray(image_slab)
                L.check(L.lib().ivx_memcpy_h2d(self.image.raw_at(lay.hb * sb), L.ptr(img), ctypes.c_size_t(img.nbytes)))
            self.comm.exchange(self.image.raw_at(lay.first_interior * sb), self.image.raw_at(0),
                               self.image.raw_at(lay.last_interior * sb), self.image.raw_at((lay.local_dz - 1) * sb), sb,
                               self.stream)
            self.sync()
            self.plane_words = self.dy * self.plan.wx
            self._recv = [DeviceBuffer(self.plane_words * 8) for _ in range(2)]
            self._votes = DeviceBuffer(64)  # int32 [0] vote travelling with the planes, [1] words my last OR gained
            self._cand = self.cand

        # the single-volume forms would take min / max and the projection over THIS rank's slices, halo included: per-rank
        # normalisation, silently different from the whole volume's.  The sharded forms are rays_global / project_global.
        def image_range(self):
            raise RuntimeError("SlabVolume.image_range: a slab does not know the whole volume's range; use rays_global "
                               "(it all-reduces min / max over the ranks' interior slices)")

        def mida(self, *a, **k):
            raise RuntimeError("SlabVolume.mida: use rays_global('mida', axis, wl, ww) -- the range and, along Z, the rays span ranks")

        def close(self):
            for b in getattr(self, "_recv", []) + [getattr(self, "_votes", None)]:
                if b is not None:
                    b.close()
            self._recv, self._votes = [], None
            super().close()

        # -- backend protocol of slab_region_grow ---------------------------------------------------------------
        def comm_span(self, name):
            return self.timer.span(name)

        def flood_run(self):
            r = ctypes.c_int(0)
            L.check(L.lib().ivx_dev_flood_run(ctypes.byref(self.plan), self._cand.ptr, self.reached.ptr,
                                              self.flood_scratch.ptr, ctypes.byref(r), self.stream), "flood_run")
            self._rounds += r.value

        def round_ptrs(self):
            nb = self.plane_words * 8
            return (self.reached.at(self.lay.first_interior * nb), self._recv[0].ptr, self.reached.at(self.lay.last_interior * nb),
                    self._recv[1].ptr, nb, self._votes.ptr, self.stream)

        def _seed_votes(self):
            """before round 1: "something happened" = the seeds -- already staged for the first all-reduce"""
            L.check(L.lib().ivx_dev_vote_set(self._votes.ptr, 1, 0, self.stream), "vote_set")  # votes[1]: or_planes adds to it

        def stage_vote(self):
            """(nothing to queue: read_votes leaves votes[0] <- votes[1] behind on the device, _seed_votes sets both words.
            Rounds 1 - 5 queued a 4-byte device copy here and two memsets above: 16 us of copy engine each)"""

        def or_planes(self):
            pd = self._recv[0].ptr if self.lay.hb else None
            pu = self._recv[1].ptr if self.lay.ht else None
            L.check(L.lib().ivx_dev_flood_or_planes_acc(ctypes.byref(self.plan), self._cand.ptr, self.reached.ptr, c64(0), pd,
                                                        c64(self.lay.local_dz - 1), pu, self.flood_scratch.ptr,
                                                        self._votes.at(4), self.stream), "flood_or_planes")

        def read_votes(self):
            """the round's only host read: both words through the pinned mailbox (no stream synchronisation, no copy-engine
            call), the next round's vote staged by the same one-thread kernel"""
            out = (ctypes.c_int32 * 2)()
            L.check(L.lib().ivx_dev_vote_read(self._votes.ptr, out, self.stream), "vote_read")
            return int(out[0]), int(out[1])

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
            self._seed_votes()
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
            if not first:  # the per-ray state arrives from the lower neighbour device to device, on this stream
                self.comm.recv(state.ptr, npix * 5 * 8, lay.rank - 1, self.stream)
            L.check(lib.ivx_dev_rays_z_slab(code, L.I16, src, c64(nint), c64(self.dy), c64(self.dx), ctypes.c_double(p0),
                                            ctypes.c_double(p1), mm.ptr, None if first else state.ptr, None if last else state.ptr,
                                            L.I16, out.ptr, status.ptr, self.stream), "rays_z_slab")
            if not last:
                self.comm.send(state.ptr, npix * 5 * 8, lay.rank + 1, self.stream)
            self.sync()
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
            sides; `marching_cubes_stitched` merges them (the vtkCleanPolyData step of join_process_surface)."""
            return super().marching_cubes_indexed(from_binary, min_value, max_value, fill_border_holes, download)

        def marching_cubes_stitched(self, from_binary=True, min_value=0, max_value=0, fill_border_holes=True, download=False):
            """The cross-slab stitch on the device (vtkAppendPolyData + vtkCleanPolyData of join_process_surface,
            surface_process.py:229-268, over the Z-slabs): this rank's indexed piece with GLOBAL vertex ids.  The vertices
            of the plane shared with the rank below exist in both pieces; they are matched by edge identity (same point
            word, same kind, same bit -- no float compares, no host arithmetic), the copies are dropped here and the faces
            point at the ids their twins have below.  Collective: one neighbour exchange of the plane's signature (32 bytes
            per point word) and one all-gather of 8 bytes per rank, both on this volume's stream.

            Returns (first global vertex id, kept vertices, triangles), or with download=True
            (first id, verts (kept, 3) float32, faces (T, 3) int32 of global ids): the ranks' arrays concatenated in rank
            order are the stitched surface.  Device buffers: self._sverts / self._faces."""
            lib, lay, st = L.lib(), self.lay, self.stream
            if from_binary is False:
                raise NotImplementedError("marching_cubes_stitched: one iso-value (from_binary) only")
            p, z0 = self._surface_params(from_binary, min_value, max_value, fill_border_holes)
            nv, nt = DeviceVolume.marching_cubes_indexed(self, from_binary, min_value, max_value, fill_border_holes, False,
                                                         params=p, z0=z0)
            if lay.world == 1:  # nothing to stitch: the piece IS the surface
                self._sverts = None
                self.stitch_counts = {"vertices": nv, "dropped_copies": 0, "global_vertices": nv}
                if download:
                    self.sync()
                    return 0, self._verts.download((nv, 3), np.float32), self._faces.download((nt, 3), np.int32)
                return 0, nv, nt
            with self.timer.span("stitch"):
                nb = ctypes.c_size_t(0)
                L.check(lib.ivx_dev_mc_stitch_sig_bytes(ctypes.byref(p), ctypes.byref(nb)))
                if getattr(self, "_sig", None) is None or self._sig[0].nbytes < nb.value:
                    self._sig = (DeviceBuffer(nb.value + 64), DeviceBuffer(nb.value + 64))
                    self._vd = DeviceBuffer(64)
                    self._vd_all = DeviceBuffer(8 * lay.world + 64)
                top, below = self._sig
                has_up, has_dn = lay.rank < lay.world - 1, lay.rank > 0
                if has_up:
                    L.check(lib.ivx_dev_mc_stitch_top_sig(ctypes.byref(p), self._mc_scratch.ptr, top.ptr, st), "stitch_top_sig")
                if lay.world > 1:
                    with self.timer.span("comm_stitch_exchange"):
                        self.comm.exchange(None, below.ptr if has_dn else None, top.ptr if has_up else None, None, nb.value, st)
                nbr = below.ptr if has_dn else None
                L.check(lib.ivx_dev_mc_stitch_match(ctypes.byref(p), self._mc_scratch.ptr, nbr, c64(nv), self._vd.ptr, st),
                        "stitch_match")
                if lay.world > 1:
                    with self.timer.span("comm_stitch_allgather"):
                        self.comm.allgather(self._vd.ptr, self._vd_all.ptr, 8, st)
                else:
                    L.check(lib.ivx_memcpy_d2d(self._vd_all.ptr, self._vd.ptr, ctypes.c_size_t(8), st))
                need = max(nv, 1) * 12
                if getattr(self, "_sverts", None) is None or self._sverts.nbytes < need:
                    if getattr(self, "_sverts", None) is not None:
                        self._sverts.close()
                    self._sverts = DeviceBuffer(int(need * 1.25) + 4096)
                L.check(lib.ivx_dev_mc_stitch_apply(ctypes.byref(p), self._mc_scratch.ptr, nbr, self._vd_all.ptr, lay.rank,
                                                    self._verts.ptr, c64(nv), self._faces.ptr, c64(nt), self._sverts.ptr, st),
                        "stitch_apply")
            self.sync()
            vd = self._vd_all.download((lay.world, 2), np.uint32).astype(np.int64)
            base = int((vd[:lay.rank, 0] - vd[:lay.rank, 1]).sum())
            kept = int(vd[lay.rank, 0] - vd[lay.rank, 1])
            self.stitch_counts = {"vertices": int(vd[lay.rank, 0]), "dropped_copies": int(vd[lay.rank, 1]),
                                  "global_vertices": int((vd[:, 0] - vd[:, 1]).sum())}
            if download:
                return base, self._sverts.download((kept, 3), np.float32), self._faces.download((nt, 3), np.int32)
            return base, kept, nt

    return SlabVolume


def __getattr__(name):  # SlabVolume needs libivx + a device; build the class lazily so the CPU tests can import us
    if name == "SlabVolume":
        cls = _make_slab_volume()
        globals()["SlabVolume"] = cls
        return cls
    raise AttributeError(name)
