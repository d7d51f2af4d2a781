"""Prototype of the ORDER-FREE statement of scikit-image's marker flood (skimage.segmentation.watershed as the reference
calls it: no mask, no compactness, no lines), checked against the serial heap flood of oracle/ivx_oracle_wssk.c with
marker ties broken by raster index (tie_mode 1).  This is the blueprint of csrc/k_wssk.hip; nothing imports it.

Serial rule: pop the smallest (image value, age); age = global push counter; a voxel takes its label when it is pushed.
Statement used here:
  C(p)    = min over paths from the markers of the largest image value on the path (markers: C = I).
  level c = {C == c}, processed in ascending c.  Inside a level:
    generation 0 = markers of value c (ordered by raster index) followed by the voxels of value c that have a neighbour
                   of lower C (ordered by the pop time of the first such neighbour to pop);
    every other voxel of the level is reached from generation 0 by steps that cost one generation INTO a voxel of value c
    and nothing INTO a voxel of value < c (those are drained at once by whoever reaches them first);
  pop time T = (G, R): G = generations counted across all levels, R = rank of the generation-0 ancestor; a voxel's label
  is the label of the neighbour with the smallest T.  Orders inside one R never matter: all those voxels share a label.
"""
import heapq
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])


def offsets_of(strct):
    s = np.asarray(strct, bool)
    if s.ndim == 2:
        s3 = np.zeros((3, 3, 3), bool)
        s3[1] = s
        s = s3
    return [(k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1) for k in range(27) if s.ravel()[k] and k != 13]


def flood_runs(image, markers, strct):
    img = np.asarray(image)
    if img.ndim == 2:
        img = img[None]
        markers = np.asarray(markers)[None]
    I = img.astype(np.int64)
    M = np.asarray(markers).astype(np.int64)
    dz, dy, dx = I.shape
    offs = offsets_of(strct)
    N = I.size
    INF = 1 << 62

    def nbrs(p):
        z, r = divmod(p, dy * dx)
        y, x = divmod(r, dx)
        for oz, oy, ox in offs:
            a, b, c = z + oz, y + oy, x + ox
            if 0 <= a < dz and 0 <= b < dy and 0 <= c < dx:
                yield (a * dy + b) * dx + c

    If, Mf = I.ravel(), M.ravel()
    # C map: Dijkstra with max as the path cost
    C = np.full(N, INF, np.int64)
    h = []
    for p in np.flatnonzero(Mf):
        C[p] = If[p]
        h.append((int(If[p]), int(p)))
    heapq.heapify(h)
    while h:
        c, p = heapq.heappop(h)
        if c != C[p]:
            continue
        for q in nbrs(p):
            if Mf[q]:
                continue
            nc = max(c, int(If[q]))
            if nc < C[q]:
                C[q] = nc
                heapq.heappush(h, (nc, q))
    tau = np.full(N, INF, np.int64)       # (G << 32) | R
    lab = np.zeros(N, np.int64)
    runlabel = []
    gbase = 1
    for c in np.unique(C[C < INF]):
        c = int(c)
        level = np.flatnonzero(C == c)
        gen0 = []
        for p in level:
            p = int(p)
            if If[p] != c:
                continue
            if Mf[p]:
                gen0.append((p, p, int(Mf[p])))
                continue
            best = None
            for q in nbrs(p):
                if C[q] < c and (best is None or tau[q] < tau[best]):
                    best = q
            if best is not None:
                gen0.append(((1 << 32) + int(tau[best]), p, int(lab[best])))
        gen0.sort()
        front = []
        for key, p, l in gen0:
            r = len(runlabel)
            runlabel.append(l)
            tau[p] = (gbase << 32) | r
            lab[p] = l
            front.append((int(tau[p]), p))
        heapq.heapify(front)
        gmax = gbase
        while front:
            t, p = heapq.heappop(front)
            if t != tau[p]:
                continue
            gmax = max(gmax, t >> 32)
            for q in nbrs(p):
                if C[q] != c or tau[q] <= t:
                    continue
                nt = t + ((1 << 32) if If[q] == c else 0)
                if nt < tau[q]:
                    tau[q] = nt
                    lab[q] = runlabel[nt & 0xFFFFFFFF]
                    heapq.heappush(front, (nt, q))
        gbase = gmax + 1
    return lab.reshape(np.asarray(image).shape).astype(np.int32)


def main():
    from oracle import oracle as O
    from scipy import ndimage
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    bad = 0
    for k in range(ncase):
        nd = 3 if k % 5 else 2
        shape = tuple(int(v) for v in rng.integers(1 if nd == 3 else 3, 9 if nd == 3 else 14, size=nd))
        levels = int(rng.choice([1, 2, 3, 6, 30, 3000]))
        img = rng.integers(0, levels, size=shape).astype(np.uint16)
        if k % 3 == 0 and min(shape) >= 3:
            img = ndimage.morphological_gradient(img, (3,) * nd)
        mk = np.zeros(shape, np.int16)
        n_mark = int(rng.integers(1, max(2, img.size // 6)))
        pos = rng.choice(img.size, size=min(n_mark, img.size), replace=False)
        mk.ravel()[pos] = rng.integers(1, 4, size=len(pos))
        st = ndimage.generate_binary_structure(nd, int(rng.integers(1, nd + 1)))
        want = O.watershed_sk(img, mk, st, 1)
        got = flood_runs(img, mk, st)
        if not np.array_equal(want, got):
            bad += 1
            if bad <= 5:
                print("MISMATCH case", k, shape, levels, "conn", st.sum() - 1, "diff", int((want != got).sum()))
    print("%d cases, %d mismatching" % (ncase, bad))


if __name__ == "__main__":
    main()
