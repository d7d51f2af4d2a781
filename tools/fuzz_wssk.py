"""Fuzz the GPU scikit-image flood (and the IFT flood) against the serial oracles: python tools/fuzz_wssk.py [cases] [seed]"""
import sys
import time

import numpy as np
from scipy import ndimage

sys.path.insert(0, ".")
from invesalius3_amd import _lib as L, watershed_process as wp  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    L.require_device()
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad_sk = bad_ift = 0
    t0 = time.time()
    for k in range(ncase):
        nd = 3 if k % 6 else 2
        hi = int(rng.choice([6, 12, 24, 44])) if nd == 3 else int(rng.choice([8, 40, 130]))
        shape = tuple(int(v) for v in rng.integers(1 if nd == 3 else 3, hi, size=nd))
        levels = int(rng.choice([1, 2, 3, 5, 17, 255, 4000, 65535]))
        img = rng.integers(0, levels, size=shape).astype(np.uint16)
        kind = k % 4
        if kind == 1 and min(shape) >= 3:
            img = ndimage.morphological_gradient(img, (3,) * nd)
        elif kind == 2:  # smooth ramps + plateaus
            img = (ndimage.uniform_filter(img.astype(np.float32), 3) // max(1, levels // 8)).astype(np.uint16)
        mk = np.zeros(shape, np.int16 if k % 2 else np.int8)
        n_mark = int(rng.integers(1, max(2, img.size // int(rng.choice([3, 10, 100, 1000])))))
        pos = rng.choice(img.size, size=min(n_mark, img.size), replace=False)
        mk.ravel()[pos] = rng.integers(1, int(rng.choice([2, 3, 9])), size=len(pos))
        if k % 9 == 0 and min(shape) >= 4:  # a brush blob
            mk[tuple(slice(0, 3) for _ in shape)] = 1
        st = ndimage.generate_binary_structure(nd, int(rng.integers(1, nd + 1)))
        got = wp.watershed(img, mk, st)
        want = O.watershed_sk(img, mk, st, 1)
        if not np.array_equal(got, want):
            bad_sk += 1
            print("SK MISMATCH case", k, shape, levels, int(st.sum()) - 1, int((got != want).sum()), flush=True)
        gi = wp.watershed_ift(img, mk, st)
        wi = O.watershed_ift_clean(img, mk, st)
        if not np.array_equal(gi, wi):
            bad_ift += 1
            print("IFT MISMATCH case", k, shape, levels, int(st.sum()) - 1, int((gi != wi).sum()), flush=True)
    print("%d cases in %.0f s: scikit-image flood mismatches %d, IFT flood mismatches %d" % (ncase, time.time() - t0, bad_sk, bad_ift))


if __name__ == "__main__":
    main()
