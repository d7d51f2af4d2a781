"""A SECOND, independent restatement of invesalius_rs/src/floodfill.rs -- plain Python, written from the Rust source's
behaviour (a LIFO walk over a structuring element's offsets, `out != fill` as the visited test, in-range seeds only) and not
from oracle/ivx_oracle.c -- against which the C restatement must agree bit for bit on random volumes.  The reference's own
golden vectors pin three tiny cases (tests/test_oracle_golden.py); the fixtures generated from the reference's Python run
over the C restatement itself, so they pin the wrappers, not the walk.  This file closes that gap from the other side: two
independent transcriptions of the walk, hundreds of random cases incl. asymmetric structuring elements (reachability is
directed then), pre-filled `out` voxels (barriers), seeds outside the range, 1 x 3 x 3 and 3 x 1 x 3 elements, the in-place
variant and fill_holes_automatically.

  generic_floodfill_threshold          floodfill.rs:96-166
  generic_floodfill_threshold_inplace  floodfill.rs:168-237
  floodfill_internal                   floodfill.rs:5-49
  fill_holes_automatically_internal    floodfill.rs:51-94
  floodfill_auto_threshold             floodfill_py.rs:12-85"""
import numpy as np
import pytest


def flood_py(data, seeds, t0, t1, fill, strct, out):
    """the LIFO walk; the result is independent of the order (plain reachability), the order is kept anyway"""
    dz, dy, dx = data.shape
    odz, ody, odx = strct.shape
    oz, oy, ox = odz // 2, ody // 2, odx // 2
    stack = []
    for (i, j, k) in seeds:
        if t0 <= data[k, j, i] <= t1:
            stack.append((i, j, k))
            out[k, j, i] = fill
    while stack:
        x, y, z = stack.pop()
        out[z, y, x] = fill
        for kk in range(odz):
            zo = z + kk - oz
            if not 0 <= zo < dz:
                continue
            for jj in range(ody):
                yo = y + jj - oy
                if not 0 <= yo < dy:
                    continue
                for ii in range(odx):
                    if not strct[kk, jj, ii]:
                        continue
                    xo = x + ii - ox
                    if not 0 <= xo < dx:
                        continue
                    if out[zo, yo, xo] != fill and t0 <= data[zo, yo, xo] <= t1:
                        out[zo, yo, xo] = fill
                        stack.append((xo, yo, zo))
    return out


def _case(rng, trial):
    shape = tuple(int(v) for v in rng.integers(1 if trial % 7 == 0 else 3, 9, 3))
    data = rng.integers(-50, 60, shape).astype(np.int16)
    t0 = int(rng.integers(-60, 0))
    t1 = int(t0 + rng.integers(0 if trial % 5 == 0 else 40, 110))
    sshape = [(3, 3, 3), (1, 3, 3), (3, 1, 3), (3, 3, 1), (1, 1, 3)][trial % 5]
    strct = (rng.random(sshape) < (0.35 if trial % 3 else 0.8)).astype(np.uint8)
    nseeds = int(rng.integers(1, 4))
    seeds = [(int(rng.integers(0, shape[2])), int(rng.integers(0, shape[1])), int(rng.integers(0, shape[0]))) for _ in range(nseeds)]
    return data, seeds, t0, t1, strct


def test_floodfill_threshold_second_restatement(oracle):
    rng = np.random.default_rng(2024)
    nonempty = 0
    for trial in range(300):
        data, seeds, t0, t1, strct = _case(rng, trial)
        fill = int(rng.integers(1, 255))
        pre = (rng.random(data.shape) < 0.1).astype(np.uint8) * (fill if trial % 2 else 7)  # barriers / other old values
        got, want = pre.copy(), pre.copy()
        oracle.floodfill_threshold(data, seeds, t0, t1, fill, strct, got)
        flood_py(data, seeds, t0, t1, fill, strct, want)
        assert np.array_equal(got, want), (trial, data.shape, strct.shape)
        nonempty += int((want != pre).sum() > 1)
    assert nonempty > 100  # the cases do flood


def test_floodfill_threshold_inplace_second_restatement(oracle):
    rng = np.random.default_rng(77)
    for trial in range(200):
        data, seeds, t0, t1, strct = _case(rng, trial)
        fill = int(rng.integers(-60, 70))  # may lie inside [t0, t1]: a filled voxel then still counts as visited (!= fill test)
        got = data.copy()
        want = data.copy()
        oracle.floodfill_threshold_inplace(got, seeds, t0, t1, fill, strct)
        flood_py(want, seeds, t0, t1, fill, strct, want)  # data and out are the same array
        assert np.array_equal(got, want), (trial, fill, t0, t1)


def test_floodfill_equal_value_second_restatement(oracle):
    rng = np.random.default_rng(5)
    six = np.zeros((3, 3, 3), np.uint8)
    six[1, 1, :] = six[1, :, 1] = six[:, 1, 1] = 1
    for trial in range(100):
        shape = tuple(int(v) for v in rng.integers(1, 8, 3))
        data = rng.integers(0, 3, shape).astype(np.uint8)
        k, j, i = (int(rng.integers(0, s)) for s in shape)
        v = int(data[k, j, i]) if trial % 4 else 9  # a value the seed does not have: the reference floods from it all the same
        got = np.zeros(shape, np.uint8)
        oracle.floodfill(data, i, j, k, v, 5, got)
        # floodfill_internal: the seed is pushed unconditionally and painted, neighbours need data == v (6-neighbourhood)
        want = np.zeros(shape, np.uint8)
        want[k, j, i] = 5
        stack = [(i, j, k)]
        while stack:
            x, y, z = stack.pop()
            for dxx, dyy, dzz in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                xo, yo, zo = x + dxx, y + dyy, z + dzz
                if 0 <= xo < shape[2] and 0 <= yo < shape[1] and 0 <= zo < shape[0] and data[zo, yo, xo] == v and want[zo, yo, xo] != 5:
                    want[zo, yo, xo] = 5
                    stack.append((xo, yo, zo))
        assert np.array_equal(got, want), trial


def test_fill_holes_automatically_second_restatement(oracle):
    from scipy import ndimage
    rng = np.random.default_rng(11)
    for trial in range(60):
        shape = tuple(int(v) for v in rng.integers(2, 10, 3))
        mask = (rng.random(shape) < 0.6).astype(np.uint8) * 255
        labels, nlabels = ndimage.label(mask == 0, output=np.uint32)
        max_size = int(rng.integers(0, 12))
        got = mask.copy()
        modified = oracle.fill_holes_automatically(got, labels, nlabels, max_size)
        sizes = np.bincount(labels.ravel(), minlength=nlabels + 1)
        small = (sizes > 0) & (sizes <= max_size)
        want = mask.copy()
        if small.any():  # note: label 0 (the mask itself) takes part like any other label, as in the Rust loop
            want[small[labels]] = 254
        assert modified == bool(small.any()) and np.array_equal(got, want), trial


def _sat_i16(f):
    """Rust's `as i16` from f32: saturating, NaN -> 0"""
    if f != f:
        return 0
    return int(max(-32768.0, min(32767.0, float(np.trunc(f)))))


def test_floodfill_auto_threshold_second_restatement(oracle):
    """floodfill_py.rs:12-85: a FIFO walk (pop_front) over the 6-neighbourhood in which the admissible range of a step depends on
    the voxel it LEAVES -- [ceil(v * (1 - p)), floor(v * (1 + p))] in f32, cast to i16 the way `as` does -- so reachability is
    directed; seeds are painted unconditionally"""
    from collections import deque
    rng = np.random.default_rng(31)
    F = np.float32
    grew = 0
    for trial in range(150):
        shape = tuple(int(v) for v in rng.integers(2, 9, 3))
        base = int(rng.integers(-300, 300))
        data = (base + rng.integers(-40, 41, shape)).astype(np.int16)
        if trial % 10 == 0:
            data.flat[int(rng.integers(0, data.size))] = 32767  # saturating casts
        p = F(rng.choice([0.05, 0.1, 0.3, 0.0]))
        fill = int(rng.integers(1, 255))
        seeds = [(int(rng.integers(0, shape[2])), int(rng.integers(0, shape[1])), int(rng.integers(0, shape[0]))) for _ in range(int(rng.integers(1, 3)))]
        pre = (rng.random(shape) < 0.05).astype(np.uint8) * fill
        got, want = pre.copy(), pre.copy()
        oracle.floodfill_auto_threshold(data, seeds, float(p), fill, got)
        q = deque()
        for (i, j, k) in seeds:
            q.append((i, j, k))
            want[k, j, i] = fill
        while q:
            x, y, z = q.popleft()
            val = F(data[z, y, x])
            t0 = _sat_i16(np.ceil(F(val * F(F(1.0) - p))))
            t1 = _sat_i16(np.floor(F(val * F(F(1.0) + p))))
            for xo, yo, zo in ((x, y, z + 1), (x, y, z - 1), (x, y + 1, z), (x, y - 1, z), (x + 1, y, z), (x - 1, y, z)):
                if 0 <= xo < shape[2] and 0 <= yo < shape[1] and 0 <= zo < shape[0] and want[zo, yo, xo] != fill:
                    if t0 <= int(data[zo, yo, xo]) <= t1:
                        want[zo, yo, xo] = fill
                        q.append((xo, yo, zo))
        assert np.array_equal(got, want), (trial, float(p))
        grew += int((want != pre).sum() > len(seeds))
    assert grew > 40
