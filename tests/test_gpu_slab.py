"""GPU: the multi-GPU slab backend (invesalius3_amd.parallel.SlabVolume) with 2 and 3 ranks driven on ONE device through
an in-process loop-back communicator (same interface as TorchComm; RCCL itself needs >= 2 GPUs).  Exercises the real HIP
path of every sharded step -- image halo, reached-plane export / OR, convergence loop, per-rank marching-cubes piece --
against the single-volume oracle.  No torch here: planes are raw device buffers (DevPlane); TorchComm wraps them in
CUDA tensors only when real RCCL traffic is needed."""
import threading

import numpy as np
import pytest
from scipy.ndimage import generate_binary_structure

from conftest import synth_volume

pytestmark = pytest.mark.gpu


class LoopbackWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.box = {}
        self.acc = [0] * world
        self.p2p = {}
        self.cv = threading.Condition()

    def comm(self, rank):
        return LoopbackComm(self, rank)


class LoopbackComm:
    def __init__(self, w, rank):
        self.w, self.rank, self.world = w, rank, w.world

    def exchange_host(self, to_down, to_up):
        return self.exchange(np.array(to_down), np.array(to_up))

    def exchange(self, to_down, to_up):
        # same process, same device: the peer reads the sender's plane in place (the sender does not touch it again
        # before the second barrier)
        w = self.w
        if self.rank > 0:
            w.box[(self.rank, "down")] = to_down
        if self.rank < self.world - 1:
            w.box[(self.rank, "up")] = to_up
        w.barrier.wait()
        from_down = w.box[(self.rank - 1, "up")] if self.rank > 0 else None
        from_up = w.box[(self.rank + 1, "down")] if self.rank < self.world - 1 else None
        w.barrier.wait()
        return from_down, from_up

    def allreduce_sum(self, value):
        w = self.w
        w.acc[self.rank] = int(value)
        w.barrier.wait()
        total = sum(w.acc)
        w.barrier.wait()
        return total

    def allreduce_array(self, a, op):
        w = self.w
        w.box[(self.rank, "arr")] = np.array(a)
        w.barrier.wait()
        parts = [w.box[(r, "arr")] for r in range(self.world)]
        out = {"max": np.maximum.reduce, "min": np.minimum.reduce, "sum": np.add.reduce}[op](parts)
        w.barrier.wait()
        return out

    def send_array(self, a, to):
        w = self.w
        with w.cv:
            w.p2p[(self.rank, to)] = np.array(a)
            w.cv.notify_all()

    def recv_array(self, shape, dtype, frm):
        w = self.w
        with w.cv:
            w.cv.wait_for(lambda: (frm, self.rank) in w.p2p, timeout=120)
            return w.p2p.pop((frm, self.rank)).reshape(shape)

    def bcast_array(self, a, shape, dtype, root):
        w = self.w
        if self.rank == root:
            w.box[("bcast", root)] = np.array(a, dtype=dtype)
        w.barrier.wait()
        out = w.box[("bcast", root)].copy()
        w.barrier.wait()
        return out

    def allgather_rows(self, a, rows_per_rank):
        w = self.w
        w.box[(self.rank, "rows")] = np.array(a)
        w.barrier.wait()
        out = np.concatenate([w.box[(r, "rows")] for r in range(self.world)], axis=0)
        w.barrier.wait()
        return out


@pytest.mark.parametrize("world,conn", [(2, 3), (3, 1)])
def test_slab_volume_matches_single_volume_oracle(ivxlib, oracle, world, conn):
    from invesalius3_amd.parallel import SlabVolume

    nz = 24
    full = synth_volume((world * nz, 48, 80), seed=78)
    t0, t1 = -850, 3071
    strct = generate_binary_structure(3, conn)
    z, y, x = np.unravel_index(int(np.argmax(full)), full.shape)
    seeds = [(int(x), int(y), int(z)), (0, 0, 0)]
    lw = LoopbackWorld(world)
    res, errs = {}, []

    def run(rank):
        try:
            vol = SlabVolume(full[rank * nz:(rank + 1) * nz], rank, world, spacing=(0.5, 0.5, 2.0), comm=lw.comm(rank))
            vol.threshold(t0, t1)
            vol.region_grow(seeds, t0, t1, strct, fill=1, select_value=254)
            tris = vol.marching_cubes(from_binary=True, download=True)
            lay = vol.lay
            proj = {(ax, op): vol.project_global(ax, op) for ax in (0, 1, 2) for op in ("max", "min", "mean")}
            piece_mesh = vol.marching_cubes_indexed(from_binary=True, download=True)
            rays = {(kind, ax): vol.rays_global(kind, ax, *par) for kind, par in (("lmip", (-300, 900)), ("mida", (300.0, 1200.0)))
                    for ax in (0, 1, 2)}
            fcm = {(tm, ax): vol.fast_countour_mip_global(2.0, ax, 300, 1200, tm) for tm in (0, 1, 2) for ax in (0, 1, 2)}
            res[rank] = dict(proj=proj, mesh=piece_mesh, rays=rays, fcm=fcm, out=vol.download_out_mask()[lay.first_interior:lay.last_interior + 1],
                             mask=vol.download_mask()[lay.first_interior:lay.last_interior + 1], tris=tris,
                             count=vol.reached_count())
            vol.close()
        except Exception as e:  # pragma: no cover
            errs.append((rank, repr(e)))
            lw.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs
    ref = np.zeros(full.shape, np.uint8)
    oracle.floodfill_threshold(full, seeds, t0, t1, 1, strct, ref)
    got = np.concatenate([res[r]["out"] for r in range(world)])
    assert np.array_equal(got, ref)
    assert sum(res[r]["count"] for r in range(world)) == int(ref.sum()) > 1000
    mask = np.zeros(tuple(s + 1 for s in full.shape), np.uint8)
    oracle.set_mask_threshold_volume(mask, full, (t0, t1))
    mask[1:, 1:, 1:][ref.astype(bool)] = 254
    assert np.array_equal(np.concatenate([res[r]["mask"] for r in range(world)]), mask[1:, 1:, 1:])
    whole = oracle.create_surface_piece(None, mask, slice(0, full.shape[0]), (0.5, 0.5, 2.0), 0, 0, True)
    cat = np.concatenate([res[r]["tris"] for r in range(world)])
    key = lambda t: np.sort(t.reshape(len(t), -1).view([("", np.float32)] * 9), axis=0)
    assert len(cat) == len(whole) and np.array_equal(key(cat), key(whole))
    # cross-slab stitch of the ranks' indexed pieces == the merged whole-volume surface
    from invesalius3_amd.parallel import stitch_piece_meshes
    sv, sf = stitch_piece_meshes([res[r]["mesh"] for r in range(world)])
    assert len(sv) == len(np.unique(whole.reshape(-1, 3), axis=0)) == len(np.unique(sv, axis=0))
    assert np.array_equal(key(sv[sf]), key(whole))
    # LMIP / MIDA of the whole volume: rays along Z are handed from slab to slab, the others are rank-local rows
    for (kind, ax), img in res[0]["rays"].items():
        oshape = tuple(s for i, s in enumerate(full.shape) if i != ax)
        want = np.zeros(oshape, np.int16)
        if kind == "lmip":
            oracle.lmip(full, ax, -300, 900, want)
        else:
            oracle.mida(full, ax, 300, 1200, want)
        for r in range(world):
            assert np.array_equal(res[r]["rays"][(kind, ax)], want), (kind, ax, r)
    # contour MIP: the contour volume needs the z neighbours across the slab boundary (halo slices)
    for (tm, ax), img in res[0]["fcm"].items():
        want = np.zeros(tuple(s for i, s in enumerate(full.shape) if i != ax), np.int16)
        oracle.fast_countour_mip(full, 2.0, ax, 300, 1200, tm, want)
        for r in range(world):
            assert np.array_equal(res[r]["fcm"][(tm, ax)], want), (tm, ax, r)
    # projections of the whole volume, identical on every rank and equal to numpy on the unsharded array
    for r in range(world):
        for (ax, op), img in res[r]["proj"].items():
            want = getattr(full, op)(axis=ax)
            assert img.dtype == want.dtype and np.array_equal(img, want), (r, ax, op)


_TORCH_PLANES = r"""
import sys, threading
import torch
torch.cuda.init()                      # torch's HIP runtime first, as in bench.py: it cannot come up after libivx's
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from scipy.ndimage import generate_binary_structure
from conftest import synth_volume
from test_gpu_slab import LoopbackWorld, LoopbackComm
from invesalius3_amd.parallel import SlabVolume
from oracle import oracle
oracle.build()


class LoopbackTorchComm(LoopbackComm):
    device = "cuda"

    def plane_buffer(self, nbytes, slot):
        bufs = self.__dict__.setdefault("_planes", {})
        if slot not in bufs or bufs[slot].numel() != nbytes:
            bufs[slot] = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        return bufs[slot]

    def exchange(self, to_down, to_up):
        if isinstance(to_down, np.ndarray) or isinstance(to_up, np.ndarray):  # the one-time image halo
            return super().exchange(to_down, to_up)
        for t in (to_down, to_up):
            assert t is None or (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8)
        # hand the peer a COPY, as a real exchange would (the sender's buffer is rewritten next round)
        cd, cu = (None if to_down is None else to_down.clone()), (None if to_up is None else to_up.clone())
        torch.cuda.current_stream().synchronize()  # as TorchComm.exchange does: the copies run on torch's stream
        return super().exchange(cd, cu)


world, nz = 3, 20
full = synth_volume((world * nz, 48, 128), seed=79)
t0, t1 = -850, 3071
strct = generate_binary_structure(3, 3)
z, y, x = np.unravel_index(int(np.argmax(full)), full.shape)
seeds = [(int(x), int(y), int(z))]
lw = LoopbackWorld(world)
res, errs = {}, []


def run(rank):
    try:
        vol = SlabVolume(full[rank * nz:(rank + 1) * nz], rank, world, comm=LoopbackTorchComm(lw, rank))
        vol.threshold(t0, t1)
        vol.region_grow(seeds, t0, t1, strct, fill=1, select_value=254)
        lay = vol.lay
        res[rank] = vol.download_out_mask()[lay.first_interior:lay.last_interior + 1]
        vol.close()
    except Exception as e:
        import traceback
        traceback.print_exc()
        errs.append((rank, repr(e)))
        lw.barrier.abort()


th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]
[t.join(timeout=300) for t in th]
assert not errs, errs
ref = np.zeros(full.shape, np.uint8)
oracle.floodfill_threshold(full, seeds, t0, t1, 1, strct, ref)
assert np.array_equal(np.concatenate([res[r] for r in range(world)]), ref)
assert ref[:nz].any() and ref[-nz:].any()  # the region really crosses both slab faces
print("torch-planes-ok")
"""


def test_slab_flood_through_cuda_tensor_planes(ivxlib):
    """3 ranks on one GPU, planes exported straight into CUDA tensors (TorchComm.plane_buffer's role) and OR-ed in from
    CUDA tensors: the plumbing RCCL traffic goes through.  In a fresh process, torch initialised first."""
    pytest.importorskip("torch")
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", _TORCH_PLANES % (os.path.dirname(here), here)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "torch-planes-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
