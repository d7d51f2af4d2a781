"""How scipy.ndimage.zoom(order=2) weighs its three taps: zoom an impulse with prefilter=False into float64 and compare,
bit for bit, with candidate closed forms (scipy's C source is not in this image).  Candidate B is the one
csrc/k_zoom.hip implements."""
import numpy as np
from scipy import ndimage
# weights probe: prefilter=False, impulse input, float64 output
n_in, n_out = 41, 29
zoomf = n_out / n_in
def W(pos):
    a = np.zeros(n_in); a[pos] = 1.0
    return ndimage.zoom(a, zoomf, output=np.float64, order=2, prefilter=False, mode='constant')
out_len = len(W(0))
z = (n_in - 1) / (out_len - 1)
print("out_len", out_len, "zoom", z)
cc = np.arange(out_len, dtype=np.float64) * z
# candidate weights
def cand_A(x):
    # x: coordinate; start = floor(x+0.5)-1 ; delta d = x - floor(x+0.5) in [-0.5,0.5)
    d = x - np.floor(x + 0.5)
    w1 = 0.75 - d * d
    w2 = 0.5 * (d + 0.5) * (d + 0.5)
    w0 = 1.0 - w1 - w2
    return w0, w1, w2
def cand_B(x):
    d = x - np.floor(x + 0.5)
    w1 = 0.75 - d * d
    y = 0.5 - d
    w0 = 0.5 * y * y
    w2 = 1.0 - w0 - w1
    return w0, w1, w2
def cand_C(x):
    d = x - np.floor(x + 0.5)
    w1 = 0.75 - d * d
    y = 0.5 + d
    w2 = 0.5 * y * y
    y2 = 0.5 - d
    w0 = 0.5*y2*y2
    return w0, w1, w2
got = np.stack([W(p) for p in range(n_in)])  # [pos, out]
for name, f in (("A", cand_A), ("B", cand_B), ("C", cand_C)):
    ok = True; bad = 0
    for k in range(out_len):
        x = cc[k]
        if x < 0 or x > n_in - 1: continue
        s = int(np.floor(x + 0.5)) - 1
        w = f(x)
        for h in range(3):
            idx = s + h
            # mirror at edges
            L = n_in; s2 = 2*L-2
            if idx < 0: idx = -idx
            elif idx >= L: idx = s2 - idx
            # accumulate (edge taps may coincide)
        exp = np.zeros(n_in)
        for h in range(3):
            idx = s + h
            if idx < 0: idx = -idx
            elif idx >= n_in: idx = 2*n_in-2 - idx
            exp[idx] += w[h]
        if not np.array_equal(exp, got[:, k]):
            bad += 1
    print(name, "mismatching outputs", bad)
print("last coordinate", repr(cc[-1]), "in range?", cc[-1] <= n_in - 1, "last col sum", got[:, -1].sum())


def prefilter_line(c):
    """ni_splines.c for order 2, modes constant / mirror: bit-identical to scipy.ndimage.spline_filter1d"""
    import math
    z = -0.171572875253809902396622551580603843  # get_filter_poles' literal, NOT sqrt(8.0) - 3.0 evaluated in double
    c = [float(v) * ((1.0 - z) * (1.0 - 1.0 / z)) for v in c]
    n = len(c)
    z_n_1 = math.pow(z, n - 1)
    c0 = c[0] + z_n_1 * c[n - 1]
    z_i = z
    for i in range(1, n - 1):
        c0 += z_i * (c[i] + z_n_1 * c[n - 1 - i])
        z_i *= z
    c[0] = c0 / (1 - z_n_1 * z_n_1)
    for i in range(1, n):
        c[i] += z * c[i - 1]
    c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1)
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])
    return np.array(c)


rng = np.random.default_rng(5)
for n in (2, 3, 5, 6, 9, 13, 40, 513):
    a = rng.integers(-1000, 3000, n).astype(np.float64)
    assert np.array_equal(prefilter_line(a), ndimage.spline_filter1d(a, 2, mode="constant")), n
print("prefilter restatement == scipy.ndimage.spline_filter1d bit for bit")
