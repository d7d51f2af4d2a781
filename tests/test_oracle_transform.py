"""A SECOND, independent restatement of invesalius_rs/src/transforms.rs + interpolation.rs (apply_view_matrix_transform: the
resampler behind a reoriented view, invesalius/data/slice_.py:862-874) in plain Python floats -- IEEE doubles, the Rust
source's operation order -- against which oracle/'s C restatement must agree bit for bit.  Upstream has no test for it;
oracle/ was cross-checked against scipy for nearest / trilinear only, so the tricubic and Lanczos-4 branches rested on one
transcription.

  coord_transform                 transforms.rs:9-55     (z, y, x, 1) * spacing -> m -> / w -> / spacing; inside test; cval clamp
  get_value                       interpolation.rs:6-35  one wrap-around step per axis
  trilinear / tricubic / lanczos  interpolation.rs:63-188"""
import math

import numpy as np
import pytest


def _get(v, x, y, z):
    dz, dy, dx = v.shape
    x = x + dx if x < 0 else (x - dx if x >= dx else x)
    y = y + dy if y < 0 else (y - dy if y >= dy else y)
    z = z + dz if z < 0 else (z - dz if z >= dz else z)
    return float(v[z, y, x])


def _cubic(p, x):
    return p[1] + 0.5 * x * (p[2] - p[0] + x * (2.0 * p[0] - 5.0 * p[1] + 4.0 * p[2] - p[3] + x * (3.0 * (p[1] - p[2]) + p[3] - p[0])))


def _bicubic(p, x, y):
    return _cubic([_cubic(p[0], y), _cubic(p[1], y), _cubic(p[2], y), _cubic(p[3], y)], x)


def _lanczos_kernel(x, a):
    if x == 0.0:
        return 1.0
    if -float(a) <= x < float(a):
        af = float(a)
        return (af * math.sin(math.pi * x) * math.sin(math.pi * (x / af))) / (math.pi * math.pi * x * x)
    return 0.0


def _trilinear(v, x, y, z):
    x0, y0, z0 = math.floor(x), math.floor(y), math.floor(z)
    x1, y1, z1 = x0 + 1, y0 + 1, z0 + 1
    xd, yd, zd = x - x0, y - y0, z - z0
    c00 = _get(v, x0, y0, z0) * (1.0 - xd) + _get(v, x1, y0, z0) * xd
    c10 = _get(v, x0, y1, z0) * (1.0 - xd) + _get(v, x1, y1, z0) * xd
    c01 = _get(v, x0, y0, z1) * (1.0 - xd) + _get(v, x1, y0, z1) * xd
    c11 = _get(v, x0, y1, z1) * (1.0 - xd) + _get(v, x1, y1, z1) * xd
    c0 = c00 * (1.0 - yd) + c10 * yd
    c1 = c01 * (1.0 - yd) + c11 * yd
    return c0 * (1.0 - zd) + c1 * zd


def _tricubic(v, x, y, z):
    xi, yi, zi = math.floor(x), math.floor(y), math.floor(z)
    p = [[[_get(v, xi + i - 1, yi + j - 1, zi + k - 1) for k in range(4)] for j in range(4)] for i in range(4)]
    arr = [_bicubic(p[i], y - yi, z - zi) for i in range(4)]
    return _cubic(arr, x - xi)


def _lanczos(v, x, y, z):
    a = 4
    xd, yd, zd = math.floor(x), math.floor(y), math.floor(z)
    xs, ys, zs = range(xd - a + 1, xd + a), range(yd - a + 1, yd + a), range(zd - a + 1, zd + a)
    temp_y = []
    for kk in zs:
        ly = 0.0
        for jj in ys:
            lx = 0.0
            for ii in xs:
                lx += _get(v, ii, jj, kk) * _lanczos_kernel(x - ii, a)
            ly += lx * _lanczos_kernel(y - jj, a)
        temp_y.append(ly)
    lz = 0.0
    for m, kk in enumerate(zs):
        lz += temp_y[m] * _lanczos_kernel(z - kk, a)
    return lz


def transform_py(volume, spacing, m, n, orientation, minterpol, cval, out_shape):
    sx, sy, sz = (float(s) for s in spacing)
    dz, dy, dx = (float(s) for s in volume.shape)
    out = np.zeros(out_shape, volume.dtype)
    M = [[float(m[r][c]) for c in range(4)] for r in range(4)]
    for cz in range(out_shape[0]):
        for cy in range(out_shape[1]):
            for cx in range(out_shape[2]):
                z, y, x = cz, cy, cx
                if orientation == "AXIAL":
                    z = n + cz
                elif orientation == "CORONAL":
                    y = n + cy
                elif orientation == "SAGITAL":
                    x = n + cx
                c = (z * sz, y * sy, x * sx, 1.0)
                nc = [M[r][0] * c[0] + M[r][1] * c[1] + M[r][2] * c[2] + M[r][3] * c[3] for r in range(4)]
                nz, ny, nx = (nc[0] / nc[3]) / sz, (nc[1] / nc[3]) / sy, (nc[2] / nc[3]) / sx
                if 0.0 <= nz < dz - 1.0 and 0.0 <= ny < dy - 1.0 and 0.0 <= nx < dx - 1.0:
                    if minterpol == 0:
                        val = volume[int(nz), int(ny), int(nx)]
                    elif minterpol == 1:
                        val = int(_trilinear(volume, nx, ny, nz))
                    else:
                        val = int(_tricubic(volume, nx, ny, nz) if minterpol == 2 else _lanczos(volume, nx, ny, nz))
                        val = cval if val < cval else val
                    out[cz, cy, cx] = val
                else:
                    out[cz, cy, cx] = cval
    return out


def _rotation(rng):
    a, b, c = rng.uniform(-0.6, 0.6, 3)
    rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    rx = np.array([[1, 0, 0], [0, math.cos(c), -math.sin(c)], [0, math.sin(c), math.cos(c)]])
    return rz @ ry @ rx


@pytest.mark.parametrize("minterpol", [0, 1, 2, 3])
def test_view_matrix_transform_second_restatement(oracle, minterpol):
    rng = np.random.default_rng(40 + minterpol)
    inside = 0
    for trial in range(6):
        shape = tuple(int(v) for v in rng.integers(6, 11, 3))
        vol = rng.integers(-900, 1000, shape).astype(np.int16)
        spacing = tuple(float(v) for v in rng.uniform(0.5, 2.0, 3))
        m = np.eye(4)
        m[:3, :3] = _rotation(rng)
        centre = np.array([shape[0] * spacing[2], shape[1] * spacing[1], shape[2] * spacing[0]]) / 2.0
        m[:3, 3] = centre - m[:3, :3] @ centre + rng.uniform(-1.0, 1.0, 3)  # rotate about the centre, nudge
        if trial == 5:
            m[3, 3] = 1.25  # a homogeneous coordinate that is not 1
        orientation = ["AXIAL", "CORONAL", "SAGITAL", "OTHER"][trial % 4]
        n = int(rng.integers(0, 3))
        oshape = [shape[0], shape[1], shape[2]]
        oshape[trial % 3] = 2 if orientation != "OTHER" else oshape[trial % 3]  # the reference resamples thin slabs
        cval = int(vol.min())
        got = np.zeros(oshape, np.int16)
        oracle.apply_view_matrix_transform(vol, spacing, m, n, orientation, minterpol, cval, got)
        want = transform_py(vol, spacing, m, n, orientation, minterpol, cval, tuple(oshape))
        assert np.array_equal(got, want), (trial, orientation, np.argwhere(got != want)[:3])
        inside += int((want != cval).sum())
    assert inside > 200  # the slabs do hit the volume
