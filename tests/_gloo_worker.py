"""Worker for tests/test_parallel_gloo.py: one rank of a world_size-N gloo job running the slab orchestration of
invesalius3_amd.parallel with a numpy/oracle backend (test infrastructure; the GPU backend is SlabVolume)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class NumpyBackend:
    """bool-array stand-in for the bit-plane kernels (flood_run = binary propagation inside `cand`) speaking the backend
    protocol of parallel.slab_region_grow with HOST pointers: planes are uint8 arrays, the votes one int32 pair."""

    def __init__(self, cand, reached, strct, lay):
        self.cand, self.reached, self.strct, self.lay = cand, reached, strct, lay
        shp = cand.shape[1:]
        self.send = [np.zeros(shp, np.uint8), np.zeros(shp, np.uint8)]
        self.recv = [np.zeros(shp, np.uint8), np.zeros(shp, np.uint8)]
        self.votes = np.array([0, 1], np.int32)  # [1] = 1: "something happened before round 1" (the seeds)
        self.reads = 0

    def flood_run(self):
        from scipy import ndimage
        if self.reached.any():
            self.reached = ndimage.binary_propagation(self.reached, structure=self.strct, mask=self.cand)

    def stage_vote(self):
        self.votes[0] = self.votes[1]

    def round_ptrs(self):
        self.send[0][:] = self.reached[self.lay.first_interior]
        self.send[1][:] = self.reached[self.lay.last_interior]
        return (self.send[0].ctypes.data, self.recv[0].ctypes.data, self.send[1].ctypes.data, self.recv[1].ctypes.data,
                self.send[0].nbytes, self.votes.ctypes.data, None)

    def or_planes(self):
        changed = 0
        for have, z, buf in ((self.lay.hb, 0, self.recv[0]), (self.lay.ht, self.lay.local_dz - 1, self.recv[1])):
            if have:
                add = buf.astype(bool) & self.cand[z] & ~self.reached[z]
                self.reached[z] |= add
                changed += int(add.sum())
        self.votes[1] = changed

    def read_votes(self):
        self.reads += 1
        return int(self.votes[0]), int(self.votes[1])


def main():
    import torch.distributed as dist
    from scipy.ndimage import generate_binary_structure

    from conftest import synth_volume
    from invesalius3_amd import parallel as par
    from oracle import oracle as orc

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    outdir, nz, conn = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = synth_volume((world * nz, 40, 70), seed=77)  # every rank can rebuild the whole volume (test only)
    t0, t1 = -850, 3071
    lay = par.slab_layout(rank, world, nz)
    lo = lay.z_global0 - lay.hb
    local = full[lo: lo + lay.local_dz]
    strct = generate_binary_structure(3, conn)
    z, y, x = np.unravel_index(int(np.argmax(full)), full.shape)
    seeds = [(int(x), int(y), int(z)), (0, 0, 0)]
    cand = (local >= t0) & (local <= t1)
    reached = np.zeros_like(cand)
    for sx, sy, sz in par.local_seeds(lay, seeds):
        if cand[sz, sy, sx]:
            reached[sz, sy, sx] = True
    from _ptr_comm import GlooPtrComm
    be = NumpyBackend(cand, reached, strct, lay)
    comm = GlooPtrComm(dist, rank, world)
    rounds = par.slab_region_grow(be, comm, lay)
    interior = be.reached[lay.first_interior: lay.last_interior + 1]
    np.save(os.path.join(outdir, "reached_%d.npy" % rank), interior)
    # halo consistency: my halo slices must equal the neighbours' interior boundary slices (checked by the parent)
    np.save(os.path.join(outdir, "halo_%d.npy" % rank), np.stack([be.reached[0], be.reached[-1]]))
    # marching cubes piece of this rank on the thresholded + selected mask
    mask_local = np.where(cand, 255, 0).astype(np.uint8)
    mask_local[be.reached] = 254
    a = par.slab_mc_args(lay)
    piece = mask_local[a["z0"]: a["z1"]]
    tris = orc.marching_cubes(piece, (0.5, 0.5, 2.0), [127.0], a["roi_start"], True, a["pad_bottom"], a["pad_top"], 0.0,
                              int(a["pad_bottom"]))
    np.save(os.path.join(outdir, "tris_%d.npy" % rank), tris)
    np.save(os.path.join(outdir, "rounds_%d.npy" % rank), np.array([rounds]))
    # projections: local reduce over the rank's own slices (numpy stands in for the HIP kernel) + the real combine
    own = local[lay.first_interior: lay.last_interior + 1]
    for ax in (0, 1, 2):
        for op in ("max", "min", "mean"):
            if op == "mean":
                partial = own.sum(axis=ax, dtype=np.int64) if ax == 0 else own.mean(axis=ax)
            else:
                partial = getattr(own, op)(axis=ax)
            img = par.slab_project_combine(partial, comm, ax, op, [nz] * world, nz * world)
            np.save(os.path.join(outdir, "proj_%d_%d_%s.npy" % (rank, ax, op)), img)
    # the hand-over primitives of the Z-ray pipeline: a token walks up the ranks, the last one's value is broadcast
    tok = np.arange(7, dtype=np.float64)
    if rank > 0:
        comm.recv(tok.ctypes.data, tok.nbytes, rank - 1, None)
    tok = tok + rank
    if rank < world - 1:
        comm.send(tok.ctypes.data, tok.nbytes, rank + 1, None)
    final = comm.bcast_array(tok if rank == world - 1 else None, (7,), np.float64, world - 1)
    assert be.reads == rounds  # ONE host read per exchange round
    np.save(os.path.join(outdir, "token_%d.npy" % rank), final)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
