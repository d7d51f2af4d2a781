import sys, ctypes
import numpy as np
sys.path.insert(0, ".")
from bench import SEED, BONE
from invesalius3_amd import _lib as L
from invesalius3_amd.device import DeviceBuffer, DeviceVolume, c64
from oracle import oracle as orc
n = 512; sl = int(sys.argv[1]) if len(sys.argv) > 1 else 381
rng = np.random.default_rng(SEED)
blobs = np.stack([rng.uniform(0.15, 0.85, 6), rng.uniform(0.15, 0.85, 6), rng.uniform(0.15, 0.85, 6), rng.uniform(0.12, 0.28, 6)], axis=1).astype(np.float32)
buf = DeviceBuffer(sl * n * n * 2)
L.check(L.lib().ivx_dev_synth_volume(buf.ptr, c64(sl), c64(n), c64(n), c64(0), c64(n), ctypes.c_uint32(SEED), L.ptr(blobs), None))
L.synchronize()
sub = buf.download((sl, n, n), np.int16)
print("range", sub.min(), sub.max(), "in range frac", float(((sub >= 226) & (sub <= 3071)).mean()))
mask = np.zeros((sl + 1, n + 1, n + 1), np.uint8)
orc.set_mask_threshold_volume(mask, sub, BONE)
rois = [slice(i * 20, i * 20 + 21) for i in range(int(round(sl / 20 + 0.5, 0))) if i * 20 < sl]
parts = [orc.create_surface_piece(None, mask, r, (1.0, 1.0, 1.0), 0, 0, True) for r in rois]
want = np.concatenate(parts)
whole = orc.create_surface_piece(None, mask, slice(0, sl), (1.0, 1.0, 1.0), 0, 0, True)
print("oracle pieces", len(want), "oracle whole", len(whole), "equal", want.shape == whole.shape and np.array_equal(want, whole))
small = DeviceVolume(sub)
small.threshold(BONE[0], BONE[1])
got = small.marching_cubes(from_binary=True, download=True)
gm = small.download_mask()
print("gpu", len(got), "mask equal", np.array_equal(gm, mask[1:, 1:, 1:]), "soup==whole", got.shape == whole.shape and np.array_equal(got, whole))
if got.shape == want.shape:
    d = np.nonzero((got != want).any(axis=(1, 2)))[0]
    print("diff tris vs pieces", len(d), d[:5], [len(p) for p in parts][:3], [len(p) for p in parts][-3:])
