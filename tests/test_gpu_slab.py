"""GPU: the multi-GPU slab backend (invesalius3_amd.parallel.SlabVolume) with 2, 3 and 8 ranks driven on ONE device through
an in-process loop-back communicator (tests/_ptr_comm.py: the pointer-level protocol of comm.RcclComm; RCCL itself refuses
two ranks on one GPU).  Exercises the real HIP path of every sharded step -- device-to-device image halo, boundary planes
sent from where they lie, OR + vote, convergence loop with one host read per round, per-rank marching-cubes piece, the
Z-ray state hand-over -- against the single-volume oracle; plus the RCCL binding itself at world size 1."""
import threading

import numpy as np
import pytest
from scipy.ndimage import generate_binary_structure

from conftest import synth_volume

pytestmark = pytest.mark.gpu


from _ptr_comm import LoopbackWorld  # noqa: E402  (N ranks as threads on ONE GPU, device-pointer protocol)


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
            vol = SlabVolume(full[rank * nz:(rank + 1) * nz], rank, world, comm=lw.comm(rank), spacing=(0.5, 0.5, 2.0))
            vol.threshold(t0, t1)
            vol.region_grow(seeds, t0, t1, strct, fill=1, select_value=254)
            tris = vol.marching_cubes(from_binary=True, download=True)
            lay = vol.lay
            proj = {(ax, op): vol.project_global(ax, op) for ax in (0, 1, 2) for op in ("max", "min", "mean")}
            piece_mesh = vol.marching_cubes_indexed(from_binary=True, download=True)
            stitched = vol.marching_cubes_stitched(from_binary=True, download=True)
            rays = {(kind, ax): vol.rays_global(kind, ax, *par) for kind, par in (("lmip", (-300, 900)), ("mida", (300.0, 1200.0)))
                    for ax in (0, 1, 2)}
            fcm = {(tm, ax): vol.fast_countour_mip_global(2.0, ax, 300, 1200, tm) for tm in (0, 1, 2) for ax in (0, 1, 2)}
            res[rank] = dict(proj=proj, mesh=piece_mesh, stitched=stitched, rays=rays, fcm=fcm, out=vol.download_out_mask()[lay.first_interior:lay.last_interior + 1],
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
    from _stitch_ref import stitch_piece_meshes
    sv, sf = stitch_piece_meshes([res[r]["mesh"] for r in range(world)])
    assert len(sv) == len(np.unique(whole.reshape(-1, 3), axis=0)) == len(np.unique(sv, axis=0))
    assert np.array_equal(key(sv[sf]), key(whole))
    # ... and the DEVICE stitch (edge identity, no float compares) gives the very same arrays, rank by rank
    bases = [res[r]["stitched"][0] for r in range(world)]
    dv = np.concatenate([res[r]["stitched"][1] for r in range(world)])
    df = np.concatenate([res[r]["stitched"][2] for r in range(world)])
    assert bases == [0] + list(np.cumsum([len(res[r]["stitched"][1]) for r in range(world)])[:-1])
    assert dv.shape == sv.shape and np.array_equal(dv.view(np.uint32), sv.view(np.uint32)), "stitched vertices"
    assert df.shape == sf.shape and np.array_equal(df, sf), "stitched faces"
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


def test_rccl_communicator_world_of_one(ivxlib):
    """The RCCL binding behind the C ABI (csrc/ivx_comm.hip, dlopen of librccl.so): unique id, ncclCommInitRank on this
    GPU, and every collective of the protocol at world size 1; a SlabVolume over it equals the plain DeviceVolume."""
    from invesalius3_amd.comm import RcclComm
    from invesalius3_amd.device import DeviceBuffer, DeviceVolume
    from invesalius3_amd.parallel import SlabVolume

    cid = RcclComm.unique_id()
    assert len(cid) == 128 and any(cid)
    comm = RcclComm(0, 1, cid)
    a = np.arange(12, dtype=np.int16).reshape(3, 4)
    assert np.array_equal(comm.allreduce_array(a, "max"), a)
    assert comm.allreduce_sum(5) == 5
    assert np.array_equal(comm.allgather_rows(a, [3]), a)
    assert np.array_equal(comm.bcast_array(a, a.shape, a.dtype, 0), a)
    buf = DeviceBuffer(4096)
    buf.upload(np.arange(1024, dtype=np.int32))
    comm.allreduce(buf.ptr, 1024, ivxlib.I32, 0, None)           # no-ops at world 1, but through librccl's entry checks
    comm.exchange_vote(buf.ptr, buf.ptr, buf.ptr, buf.ptr, 64, buf.ptr, 1, None)
    comm.sync()
    assert np.array_equal(buf.download((1024,), np.int32), np.arange(1024, dtype=np.int32))
    img = synth_volume((20, 32, 64), seed=3)
    strct = generate_binary_structure(3, 3)
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    sv = SlabVolume(img, 0, 1, comm=comm)
    dv = DeviceVolume(img)
    for v in (sv, dv):
        v.threshold(-850, 3071)
        v.region_grow([(int(x), int(y), int(z))], -850, 3071, strct, fill=1, select_value=254)
    assert np.array_equal(sv.download_mask(), dv.download_mask()) and sv.reached_count() == dv.reached_count() > 100
    sv.close()
    dv.close()
    comm.close()
