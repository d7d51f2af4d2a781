import sys, time
import numpy as np
from scipy import ndimage
sys.path.insert(0, ".")
from bench import synth_v512
from invesalius3_amd import watershed_process as wp
img = synth_v512((64, 512, 512))[32]
grad = wp.cost_image(img, True, 300, 400, (3, 3))
mk = np.zeros(img.shape, np.int16)
y, x = np.unravel_index(int(np.argmax(img)), img.shape)
mk[y - 3:y + 4, x - 3:x + 4] = 1
mk[:6, :6] = 2
st = ndimage.generate_binary_structure(2, 1)
wp.watershed(grad, mk, st)
for _ in range(3):
    t = time.perf_counter(); lab, s = wp.watershed(grad, mk, st, want_stats=True); dt = time.perf_counter() - t
    print("2-D 512x512 sk flood: %.1f ms" % (dt * 1e3), {k: s[k] for k in ("levels", "generations", "frontier_launches", "small_level_runs", "tile_rounds")})
t = time.perf_counter(); li = wp.watershed_ift(grad, mk.astype(np.int16), st); print("2-D IFT: %.1f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); li = wp.watershed_ift(grad, mk.astype(np.int16), st); print("2-D IFT: %.1f ms" % ((time.perf_counter() - t) * 1e3))
from oracle import oracle as O
t = time.perf_counter(); o = O.watershed_sk(grad, mk, st, 0); print("serial heap flood: %.1f ms, equal %s" % ((time.perf_counter() - t) * 1e3, np.array_equal(o, lab)))
