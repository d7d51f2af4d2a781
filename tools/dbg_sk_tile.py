import sys, faulthandler
faulthandler.enable()
import numpy as np
from scipy import ndimage
sys.path.insert(0, ".")
from invesalius3_amd import watershed_process as wp
from oracle import oracle as O
rng = np.random.default_rng(3)
for k, shape in enumerate([(4, 5, 6), (9, 17, 33), (1, 20, 20), (20, 40, 40), (33, 64, 72)]):
    img = rng.integers(0, 3, size=shape).astype(np.uint16)
    mk = np.zeros(shape, np.int16)
    pos = rng.choice(img.size, size=5, replace=False)
    mk.ravel()[pos] = rng.integers(1, 3, size=5)
    for conn in (1, 3):
        st = ndimage.generate_binary_structure(3, conn)
        print("case", k, shape, conn, flush=True)
        got, stats = wp.watershed(img, mk, st, want_stats=True)
        want = O.watershed_sk(img, mk, st, 1)
        print("  equal", np.array_equal(got, want), stats["tile_rounds"], stats["levels"], flush=True)
