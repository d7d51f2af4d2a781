"""How many generation-0 voxels of a level have ALL their lower-cost neighbours in the level right below (their key cannot be
known before that level has been flooded)?  Decides whether a level's generation 0 can be keyed and sorted while the level
below is still running.  GUI-default watershed on the bench volume; python tools/late_fraction.py [edge=512]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from bench import synth_v512  # noqa: E402
from invesalius3_amd import _lib as L, watershed_process as wp  # noqa: E402
from scipy.ndimage import generate_binary_structure  # noqa: E402
from tools.bench_wsift import markers_for  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L.require_device()
img = synth_v512((n, n, n))
mk = markers_for(img).astype(np.int16)
grad = wp.cost_image(img, True, 300, 400, (3, 3, 3))
_, cost = wp.watershed(grad, mk, generate_binary_structure(3, 1), want_cost=True)
C = cost.astype(np.int32)
levels = np.unique(C)
prev = np.full(65537, -1, np.int32)
prev[levels[1:]] = levels[:-1]
INF = 1 << 20
minlow = np.full(C.shape, INF, np.int32)
for ax in range(3):
    for sh in (1, -1):
        q = np.roll(C, sh, axis=ax)
        sl = [slice(None)] * 3
        sl[ax] = slice(0, 1) if sh == 1 else slice(-1, None)
        q[tuple(sl)] = INF
        lower = q < C
        minlow = np.where(lower & (q < minlow), q, minlow)
gen0 = (grad.astype(np.int32) == C) & (minlow < INF) & (mk == 0)
late = gen0 & (minlow == prev[C])
g0 = np.bincount(C[gen0], minlength=65536)
lt = np.bincount(C[late], minlength=65536)
big = np.nonzero(g0 > 4096)[0]
print(json.dumps({"n": n, "gen0_total": int(gen0.sum()), "late_total": int(late.sum()), "levels_over_4096": int(len(big)),
                  "late_per_big_level_median": float(np.median(lt[big])), "late_per_big_level_max": int(lt[big].max()),
                  "late_per_big_level_p90": float(np.percentile(lt[big], 90)),
                  "gen0_per_big_level_median": float(np.median(g0[big])),
                  "first_levels": [(int(c), int(g0[c]), int(lt[c])) for c in big[:12]]}))
