import sys, threading
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from scipy.ndimage import generate_binary_structure
from _ptr_comm import LoopbackWorld
from bench import synth_v512
from invesalius3_amd.device import DeviceVolume
from invesalius3_amd.parallel import SlabVolume
from oracle import oracle as orc
S26 = generate_binary_structure(3, 3); BONE = (226, 3071)
world, nz, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
full = synth_v512((world * nz, W, W), seed=7)
z, y, x = np.unravel_index(int(np.argmax(full)), full.shape)
seeds = [(int(x), int(y), int(z))]
ref = np.zeros(full.shape, np.uint8)
orc.floodfill_threshold(full, seeds, BONE[0], BONE[1], 1, S26, ref)
print("oracle", int(ref.sum()), flush=True)
one = DeviceVolume(full)
one.threshold(*BONE)
one.region_grow(seeds, BONE[0], BONE[1], S26, fill=1, select_value=254)
o1 = one.download_out_mask()
print("single", int(o1.sum()), "diff vs oracle", int((o1 != ref).sum()), flush=True)
one.close()
lw = LoopbackWorld(world); res = {}; errs = []
def run(rank):
    try:
        vol = SlabVolume(full[rank * nz:(rank + 1) * nz], rank, world, comm=lw.comm(rank))
        vol.threshold(*BONE)
        vol.region_grow(seeds, BONE[0], BONE[1], S26, fill=1, select_value=254)
        lay = vol.lay
        res[rank] = vol.download_out_mask()[lay.first_interior:lay.last_interior + 1]
        vol.close()
    except Exception:
        import traceback; errs.append(traceback.format_exc()); lw.barrier.abort()
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
print(errs)
got = np.concatenate([res[r] for r in range(world)])
d = got != ref
print("sharded", int(got.sum()), "diff vs oracle", int(d.sum()), "collectives", lw.collectives)
if d.any():
    zz = np.nonzero(d.any(axis=(1, 2)))[0]
    print("slices with diffs", zz[:50], "per rank", [int(d[r*nz:(r+1)*nz].sum()) for r in range(world)])
    print("missing (ref=1,got=0)", int((ref.astype(bool) & ~got.astype(bool)).sum()), "extra", int((~ref.astype(bool) & got.astype(bool)).sum()))
