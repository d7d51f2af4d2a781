import sys
import numpy as np
from scipy import ndimage
sys.path.insert(0, 'tools')
from proto_ws_zones import ift_zones, offsets_of


def serial_clean(img, mk, strct):
    """LIFO bucket flood without scipy's linked-list quirks; records pop order."""
    shape = img.shape
    I = img.ravel().astype(np.int64); M = mk.ravel().astype(np.int64); N = I.size
    offs = offsets_of(shape if img.ndim == 3 else (1,) + shape, strct)
    maxv = int(I.max())
    cost = np.full(N, maxv + 1, np.int64)
    out = M.copy()
    stacks = [[] for _ in range(maxv + 2)]
    stamp = np.zeros(N, np.int64)  # validity stamp for lazy deletion
    cnt = 0
    for j in range(N):
        if M[j] != 0:
            cost[j] = 0
            cnt += 1; stamp[j] = cnt
            stacks[0].append((j, cnt))
    done = np.zeros(N, bool)
    T = np.full(N, -1, np.int64); t = 0
    par = np.full(N, -1, np.int64)
    for c in range(maxv + 1):
        st = stacks[c]
        while st:
            v, s = st.pop()
            if done[v] or stamp[v] != s or cost[v] != c:
                continue
            done[v] = True; T[v] = t; t += 1
            for o in offs:
                p = v + o
                if 0 <= p < N and not done[p]:
                    m = max(cost[v], abs(int(I[p]) - int(I[v])))
                    if m < cost[p]:
                        cost[p] = m; out[p] = out[v]; par[p] = v
                        cnt += 1; stamp[p] = cnt
                        stacks[m].append((p, cnt))
    return out.reshape(shape), T, par, cost


def gen(seed, it_target):
    rng = np.random.default_rng(seed)
    for it in range(it_target + 1):
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
        if rng.random() < 0.3 and img.size > 8:
            sl = tuple(slice(0, max(1, s // 2)) for s in shape)
            mk[sl] = 2
        s = ndimage.generate_binary_structure(nd, conn)
    return img, mk, s


for seed, it in ((2, 1), (3, 5), (5, 124), (4, 375)):
    img, mk, s = gen(seed, it)
    exp = ndimage.watershed_ift(img, mk, s)
    cl, T, par, cost = serial_clean(img, mk, s)
    got, info = ift_zones(img, mk, s)
    print(seed, it, img.shape, "scipy vs clean:", int((exp != cl).sum()), " zones vs clean:", int((got != cl).sum()),
          " zones vs scipy:", int((got != exp).sum()))
