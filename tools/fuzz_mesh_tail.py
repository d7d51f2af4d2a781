"""fuzz: hole filling + point normals kernels (csrc/k_meshtail.hip) against the rules in plain Python (tests/_mesh_tail_ref.py) on
random marching-cubes surfaces -- random shapes, thresholds, border filling, hole sizes, feature angles -- array for array.
    python tools/fuzz_mesh_tail.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _mesh_tail_ref as ref
from conftest import synth_volume
from invesalius3_amd import surface_process as sp

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, bad = time.time(), 0
for c in range(cases):
    shape = tuple(int(v) for v in rng.integers(6, 28, 3))
    img = synth_volume(shape, seed=int(rng.integers(1 << 30)))
    lo = int(rng.integers(-200, 900))
    mask = np.zeros(tuple(s + 1 for s in shape), np.uint8)
    mask[1:, 1:, 1:] = np.where(img >= lo, 255, 0)
    closed = bool(rng.integers(2))
    v, f, _ = sp.join_process_volume(None, mask, tuple(rng.uniform(0.4, 2.0, 3)), 0, 0, True, fill_border_holes=closed)
    if not len(f):
        continue
    if rng.integers(3) == 0:  # knock random triangles out: holes of every size, pinch points, tangles
        f = f[rng.random(len(f)) > 0.03]
    hole = float(rng.choice([0.5, 2.0, 8.0, 300.0]))
    gv, gf, gn = sp.fill_holes(v, f, hole)
    rv, rf, rn = ref.fill_holes(v, f, hole)
    ok = gn == rn and np.array_equal(gf, rf) and np.array_equal(gv.view(np.uint32), rv.view(np.uint32))
    ang, split = float(rng.choice([20.0, 80.0, 140.0])), bool(rng.integers(4))
    g = sp.point_normals(gv, gf, ang, split, True)
    r = ref.point_normals(rv, rf, ang, split, True)
    ok = ok and np.array_equal(g[1], r[1]) and all(g[k].shape == r[k].shape and np.array_equal(g[k].view(np.uint32), r[k].view(np.uint32)) for k in (0, 2, 3))
    bad += not ok
    if not ok:
        print("MISMATCH case", c, shape, lo, closed, hole, ang, split, len(f))
print("fuzz_mesh_tail: %d cases, %d mismatches, %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
