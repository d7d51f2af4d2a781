"""Prototype (CPU, pure Python): the ZONE formulation of scipy.ndimage.watershed_ift for positive markers.

Checks the theory the HIP kernels are built on, against live scipy:
  cost C = minimax |dI| path cost from the markers (unique);
  a node of final cost c is an ENTRY if a neighbour v has C(v) < c and |I(v)-I(p)| == c (markers: level-0 entries);
  non-entry nodes of equal cost c linked by arcs <= c form ZONES; a zone goes, whole, to the adjacent entry
  (arc <= c) that the serial stack pops first = the one pushed LAST = the one whose parent was popped last;
  pop times are only ever compared across different labels, so a coarse time stamp tau per (level, key rank)
  is enough: tau(level c node) = base_c + dense rank (descending key), key(entry) = min tau over its eligible parents.
"""
import heapq
import sys

import numpy as np
from scipy import ndimage


def offsets_of(shape, strct):
    st = np.asarray(strct).astype(bool)
    dims = list(shape)
    while st.ndim < 3:
        st = st[np.newaxis]
    if st.shape[0] == 1 and len(dims) == 3:
        pass
    strides = [dims[1] * dims[2], dims[2], 1] if len(dims) == 3 else [0, dims[1], 1]
    offs = []
    cz0 = st.shape[0] // 2
    for kz in range(st.shape[0]):
        for ky in range(3):
            for kx in range(3):
                if st[kz, ky, kx]:
                    o = (kz - cz0) * strides[0] + (ky - 1) * strides[1] + (kx - 1)
                    if o != 0:
                        offs.append(o)
    return offs


def ift_zones(img, markers, strct):
    shape = img.shape
    I = img.ravel().astype(np.int64)
    M = markers.ravel().astype(np.int64)
    N = I.size
    offs = offsets_of(shape if img.ndim == 3 else (1,) + shape, strct)
    assert (M >= 0).all()
    INF = 1 << 40
    # 1. minimax cost (any order: unique)
    C = np.full(N, INF, np.int64)
    heap = []
    for i in np.flatnonzero(M):
        C[i] = 0
        heap.append((0, int(i)))
    heapq.heapify(heap)
    done = np.zeros(N, bool)
    while heap:
        c, v = heapq.heappop(heap)
        if done[v]:
            continue
        done[v] = True
        for o in offs:
            p = v + o
            if 0 <= p < N and not done[p]:
                m = max(c, abs(int(I[p]) - int(I[v])))
                if m < C[p]:
                    C[p] = m
                    heapq.heappush(heap, (m, p))
    reach = C < INF
    # 2. entries
    E = np.zeros(N, bool)
    for p in range(N):
        if not reach[p]:
            continue
        if M[p] != 0:
            E[p] = True
            continue
        c = C[p]
        for o in offs:
            v = p + o
            if 0 <= v < N and C[v] < c and abs(int(I[p]) - int(I[v])) == c:
                E[p] = True
                break
    # 3. zones = components of non-entries (same cost, arc <= cost)
    parent = np.arange(N)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for p in range(N):
        if not reach[p] or E[p]:
            continue
        for o in offs:
            q = p + o
            if 0 <= q < N and reach[q] and not E[q] and C[q] == C[p] and abs(int(I[p]) - int(I[q])) <= C[p]:
                a, b = find(p), find(q)
                if a != b:
                    parent[max(a, b)] = min(a, b)
    comp = np.array([find(p) for p in range(N)])
    # 4. level chain
    tau = np.full(N, -1, np.int64)      # per node (entries) / per zone root (non-entries)
    lab_of_tau = []

    def tau_of(v):
        return tau[v] if E[v] else tau[comp[v]]

    levels = sorted(set(int(c) for c in C[reach]))
    base = 0
    for c in levels:
        ents = [p for p in range(N) if reach[p] and E[p] and C[p] == c]
        key = {}
        if c == 0:
            mk = [p for p in ents if M[p] != 0]
            # raster order; the LAST marker is popped first -> key = raster rank
            for r, p in enumerate(sorted(mk)):
                key[p] = ('m', r)
            # level-0 non-marker entries cannot exist (no lower level)
            assert len(mk) == len(ents)
            ks = sorted(set(key.values()), reverse=True)
            rank = {k: i for i, k in enumerate(ks)}
            klabel = {('m', r): int(M[p]) for r, p in enumerate(sorted(mk))}
        else:
            for p in ents:
                best = None
                for o in offs:
                    v = p + o
                    if 0 <= v < N and C[v] < c and abs(int(I[p]) - int(I[v])) == c:
                        t = tau_of(v)
                        assert t >= 0
                        if best is None or t < best:
                            best = t
                key[p] = best
            ks = sorted(set(key.values()), reverse=True)
            rank = {k: i for i, k in enumerate(ks)}
            klabel = {k: lab_of_tau[k] for k in ks}
        # merge runs of consecutive keys that carry the same label into ONE time stamp (only cross-label order matters)
        cls = {}
        ncls = 0
        prev = None
        for k in ks:
            if prev is not None and klabel[k] != prev:
                ncls += 1
            cls[k] = ncls
            prev = klabel[k]
        ncls += 1
        for k in ks:
            rank[k] = cls[k]
        labs = [None] * ncls
        for k in ks:
            labs[cls[k]] = klabel[k]
        lab_of_tau.extend(labs)
        ks = list(range(ncls))
        for p in ents:
            tau[p] = base + rank[key[p]]
        # zones adjacent to entries take the earliest (smallest tau)
        for p in ents:
            for o in offs:
                q = p + o
                if 0 <= q < N and reach[q] and not E[q] and C[q] == c and abs(int(I[p]) - int(I[q])) <= c:
                    r = comp[q]
                    if tau[r] < 0 or tau[p] < tau[r]:
                        tau[r] = tau[p]
        base += len(ks)
    out = np.zeros(N, markers.dtype)
    for p in range(N):
        if reach[p]:
            out[p] = lab_of_tau[tau_of(p)]
    return out.reshape(shape), dict(levels=len(levels), taus=base)


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    nfail = 0
    ncase = 0
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
        nd = rng.choice([2, 3])
        if nd == 3:
            shape = tuple(int(v) for v in rng.integers(1, 9, 3))
        else:
            shape = tuple(int(v) for v in rng.integers(1, 14, 2))
        conn = int(rng.integers(1, nd + 1))
        hi = int(rng.choice([2, 4, 10, 60, 3000]))
        img = rng.integers(0, hi, shape).astype(np.uint16)
        if rng.random() < 0.5:
            img[rng.random(shape) < 0.4] = 0
        if rng.random() < 0.3:
            img = ndimage.uniform_filter(img.astype(float), 3).astype(np.uint16)
        mk = np.zeros(shape, np.int16)
        nm = int(rng.integers(1, 8))
        idx = rng.integers(0, img.size, nm)
        mk.ravel()[idx] = rng.choice(np.array([1, 2, 3], np.int16), nm)
        if rng.random() < 0.3 and img.size > 8:  # blobs of markers
            sl = tuple(slice(0, max(1, s // 2)) for s in shape)
            mk[sl] = 2
        s = ndimage.generate_binary_structure(nd, conn)
        exp = ndimage.watershed_ift(img, mk, s)
        got, info = ift_zones(img, mk, s)
        ncase += 1
        if not np.array_equal(got, exp):
            nfail += 1
            print("MISMATCH case", it, shape, conn, hi, "diff voxels", int((got != exp).sum()), info)
    print("cases", ncase, "failures", nfail)


if __name__ == "__main__":
    main()
