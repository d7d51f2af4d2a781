"""CPU, world_size 2 and 3 over gloo: the Z-slab orchestration (invesalius3_amd.parallel) reproduces the
single-volume result -- region growing across slab boundaries (halo bit-plane exchange to the global fix-point) and
the marching-cubes piece decomposition (roi / pad / one overlap slice, surface.py:1362-1380)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from scipy.ndimage import generate_binary_structure

from conftest import synth_volume

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,conn", [(2, 3), (3, 1)])
def test_slab_region_grow_and_mc_match_single_volume(tmp_path, oracle, world, conn):
    nz = 12
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), str(tmp_path),
                                       str(nz), str(conn)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)

    full = synth_volume((world * nz, 40, 70), seed=77)
    t0, t1 = -850, 3071
    strct = generate_binary_structure(3, conn)
    z, y, x = np.unravel_index(int(np.argmax(full)), full.shape)
    ref = np.zeros(full.shape, np.uint8)
    oracle.floodfill_threshold(full, [(int(x), int(y), int(z)), (0, 0, 0)], t0, t1, 1, strct, ref)
    got = np.concatenate([np.load(tmp_path / ("reached_%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got, ref.astype(bool))
    assert ref.sum() > 1000 and any(int(np.load(tmp_path / ("rounds_%d.npy" % r))[0]) >= 2 for r in range(world))
    # halos converged to the neighbours' boundary slices
    for r in range(world - 1):
        top_halo = np.load(tmp_path / ("halo_%d.npy" % r))[1]
        assert np.array_equal(top_halo, np.load(tmp_path / ("reached_%d.npy" % (r + 1)))[0])

    # marching cubes: concatenated rank pieces == whole-volume soup (as multisets of triangles)
    mask = np.zeros(tuple(s + 1 for s in full.shape), np.uint8)
    mask[1:, 1:, 1:] = np.where((full >= t0) & (full <= t1), 255, 0)
    mask[1:, 1:, 1:][ref.astype(bool)] = 254
    whole = oracle.create_surface_piece(None, mask, slice(0, full.shape[0]), (0.5, 0.5, 2.0), 0, 0, True)
    cat = np.concatenate([np.load(tmp_path / ("tris_%d.npy" % r)) for r in range(world)])
    key = lambda t: np.sort(t.reshape(len(t), -1).view([("", np.float32)] * 9), axis=0)
    assert len(cat) == len(whole) and np.array_equal(key(cat), key(whole))
    # ray-state hand-over primitives (send / recv up the ranks, broadcast from the last)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("token_%d.npy" % r)), np.arange(7.0) + sum(range(world)))
    # sharded MaxIP / MinIP / MeanIP: every rank ends with numpy's result on the whole volume
    for r in range(world):
        for ax in (0, 1, 2):
            for op in ("max", "min", "mean"):
                img = np.load(tmp_path / ("proj_%d_%d_%s.npy" % (r, ax, op)))
                want = getattr(full, op)(axis=ax)
                assert img.dtype == want.dtype and np.array_equal(img, want), (r, ax, op)


def test_layout_helpers():
    from invesalius3_amd import parallel as par
    l0, l1, l2 = (par.slab_layout(r, 3, 10) for r in range(3))
    assert (l0.hb, l0.ht, l0.local_dz, l0.first_interior, l0.last_interior) == (0, 1, 11, 0, 9)
    assert (l1.hb, l1.ht, l1.local_dz, l1.first_interior, l1.last_interior, l1.z_global0) == (1, 1, 12, 1, 10, 10)
    assert (l2.hb, l2.ht, l2.local_dz) == (1, 0, 11)
    assert par.slab_mc_args(l0) == dict(z0=0, z1=11, roi_start=0, pad_bottom=True, pad_top=False)
    assert par.slab_mc_args(l1) == dict(z0=1, z1=12, roi_start=10, pad_bottom=False, pad_top=False)
    assert par.slab_mc_args(l2) == dict(z0=1, z1=11, roi_start=20, pad_bottom=False, pad_top=True)
    assert par.local_seeds(l1, [(3, 4, 9), (3, 4, 10), (3, 4, 20), (3, 4, 21)]) == [(3, 4, 0), (3, 4, 1), (3, 4, 11)]
    one = par.slab_layout(0, 1, 7)
    assert (one.hb, one.ht) == (0, 0) and par.slab_mc_args(one)["pad_top"]
