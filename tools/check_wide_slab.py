"""one-off: the sharded path on 2048-wide slices (BASELINE configs[3] geometry, few slices) against the oracle"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from scipy.ndimage import generate_binary_structure
from test_gpu_slab import LoopbackWorld
import threading
from _stitch_ref import stitch_piece_meshes
    from invesalius3_amd.parallel import SlabVolume
from oracle import oracle
oracle.build()
world, nz, ny, nx = 2, 6, 2048, 2048
rng = np.random.default_rng(0)
zz, yy, xx = np.meshgrid(np.linspace(-1, 1, world * nz, dtype=np.float32), np.linspace(-1, 1, ny, dtype=np.float32),
                         np.linspace(-1, 1, nx, dtype=np.float32), indexing="ij")
f = 1500.0 * np.exp(-(yy ** 2 + xx ** 2) * 3.0) * (1.0 + 0.2 * zz) + 200 * np.sin(40 * xx) * np.cos(31 * yy)
f += rng.standard_normal(f.shape).astype(np.float32) * 30
full = np.clip(f - 500, -1024, 3071).astype(np.int16)
del f, zz, yy, xx
t0, t1 = 226, 3071
strct = generate_binary_structure(3, 3)
z, y, x = np.unravel_index(int(np.argmax(full)), full.shape)
seeds = [(int(x), int(y), int(z))]
lw = LoopbackWorld(world); res = {}; errs = []
def run(rank):
    try:
        vol = SlabVolume(full[rank * nz:(rank + 1) * nz], rank, world, spacing=(0.5, 0.5, 1.0), comm=lw.comm(rank))
        vol.threshold(t0, t1); vol.region_grow(seeds, t0, t1, strct, fill=1, select_value=254)
        tris = vol.marching_cubes(from_binary=True, download=True)
        lay = vol.lay
        res[rank] = dict(out=vol.download_out_mask()[lay.first_interior:lay.last_interior + 1], tris=tris,
                         mask=vol.download_mask()[lay.first_interior:lay.last_interior + 1])
        vol.close()
    except Exception as e:
        import traceback; traceback.print_exc(); errs.append(repr(e)); lw.barrier.abort()
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join(600) for t in th]
assert not errs, errs
ref = np.zeros(full.shape, np.uint8)
oracle.floodfill_threshold(full, seeds, t0, t1, 1, strct, ref)
got = np.concatenate([res[r]["out"] for r in range(world)])
print("flood equal:", np.array_equal(got, ref), int(ref.sum()))
mask = np.zeros(tuple(s + 1 for s in full.shape), np.uint8)
oracle.set_mask_threshold_volume(mask, full, (t0, t1))
mask[1:, 1:, 1:][ref.astype(bool)] = 254
print("mask equal:", np.array_equal(np.concatenate([res[r]["mask"] for r in range(world)]), mask[1:, 1:, 1:]))
t = time.time()
whole = oracle.create_surface_piece(None, mask, slice(0, full.shape[0]), (0.5, 0.5, 1.0), 0, 0, True)
cat = np.concatenate([res[r]["tris"] for r in range(world)])
key = lambda a: np.sort(a.reshape(len(a), -1).view([("", np.float32)] * 9), axis=0)
print("tris:", len(cat), len(whole), "equal:", len(cat) == len(whole) and np.array_equal(key(cat), key(whole)), "oracle s", round(time.time() - t, 1))
