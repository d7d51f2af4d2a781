"""CPU checks of the mask-editing oracle (oracle/ivx_oracle_edit.c) against independent numpy / pure-Python
transcriptions of the Rust sources on small inputs."""
import numpy as np
import pytest


def _py_polygon(shape, pts):
    w, h = shape
    out = np.zeros((w, h), bool)
    n = len(pts)
    for r in range(w):
        for c in range(h):
            inside, j = False, n - 1
            for i in range(n):
                xi, yi, xj, yj = pts[i][0], pts[i][1], pts[j][0], pts[j][1]
                if ((yi > c) != (yj > c)) and (r < (xj - xi) * (c - yi) / (yj - yi) + xi):
                    inside = not inside
                j = i
            out[r, c] = inside
    return out


def test_polygon2mask_matches_transcription_and_skimage_convention(oracle):
    pts = np.array([[2.5, 1.0], [17.2, 3.3], [12.0, 14.9], [6.1, 9.0], [1.0, 12.5]])
    got = oracle.polygon2mask((20, 16), pts)
    assert got.shape == (20, 16) and got.dtype == np.bool_
    # bounding box in the reference only skips work: the full scan gives the same mask
    assert np.array_equal(got, _py_polygon((20, 16), pts))
    assert got.sum() > 40
    assert not oracle.polygon2mask((20, 16), np.zeros((0, 2))).any()
    # polygon partly off-screen is clipped, not wrapped
    off = oracle.polygon2mask((8, 8), np.array([[-5.0, -5.0], [30.0, -5.0], [30.0, 30.0], [-5.0, 30.0]]))
    assert off.all()


def test_brush_mask_modes(oracle):
    out = np.full((9, 10, 11), 200, np.uint8)
    oracle.brush_mask(out, None, (1.0, 1.0, 1.0), (5.0, 5.0, 4.0), 2.0, 1)
    z, y, x = np.mgrid[:9, :10, :11]
    ball = (x - 5.0) ** 2 + (y - 5.0) ** 2 + (z - 4.0) ** 2 <= 4.0
    assert np.array_equal(out == 0, ball)
    orig = np.where(np.arange(11)[None, None, :] % 2 == 0, 77, 0).astype(np.uint8) * np.ones((9, 10, 1), np.uint8)
    oracle.brush_mask(out, orig, (1.0, 1.0, 1.0), (5.0, 5.0, 4.0), 2.0, 0)
    assert np.array_equal(out[ball & (orig > 0)], orig[ball & (orig > 0)]) and (out[ball & (orig == 0)] == 0).all()
    out2 = np.zeros((4, 4, 4), np.uint8)
    oracle.brush_mask(out2, None, (0.5, 0.5, 2.0), (0.75, 0.75, 2.0), 0.6, 0)
    assert out2.sum() == 255 * int(out2.astype(bool).sum()) and out2.any()
    before = out2.copy()
    oracle.brush_mask(out2, None, (0.5, 0.5, 2.0), (0.75, 0.75, 2.0), 0.6, 7)  # unknown mode: untouched
    assert np.array_equal(out2, before)


def test_mask_cut_orthographic_case(oracle):
    out = np.full((6, 8, 10), 255, np.uint8)
    out[0] = 100  # <= 127: never touched
    m = np.eye(4)
    m[0, 0], m[1, 1] = 2 / 9.0, 2 / 7.0  # x in [0,9] -> [-1,1] after the -1 shift below
    m[0, 3], m[1, 3] = -1.0, -1.0
    mv = np.eye(4)
    mask = np.zeros((8, 10), bool)
    mask[2:5, 3:7] = True
    want = out.copy()
    oracle.mask_cut(out, 1.0, 1.0, 1.0, 1e9, mask, m, mv, 1)
    z, y, x = np.mgrid[:6, :8, :10].astype(np.float64)
    q0 = ((m[0, 0] * x + m[0, 1] * y) + m[0, 2] * z) + m[0, 3] * 1.0
    q1 = ((m[1, 0] * x + m[1, 1] * y) + m[1, 2] * z) + m[1, 3] * 1.0
    px, py = (q0 / 2.0 + 0.5) * 9.0, (q1 / 2.0 + 0.5) * 7.0
    on = (px >= 0) & (px < 10) & (py >= 0) & (py < 8)
    hit = np.zeros(out.shape, bool)
    hit[on] = mask[py[on].astype(np.int64), px[on].astype(np.int64)]
    want[hit & (want > 127)] = 0
    assert np.array_equal(out, want)
    assert (out[1:, 3, 4:6] == 0).all() and (out[0] == 100).all() and (out[1:, 6:, :] == 255).all()
    # max_depth: camera distance is |p| with mv = identity
    out = np.full((6, 8, 10), 255, np.uint8)
    oracle.mask_cut(out, 1.0, 1.0, 1.0, 4.0, np.ones((8, 10), bool), m, mv, 1)
    z, y, x = np.mgrid[:6, :8, :10]
    assert np.array_equal(out == 0, np.sqrt(x * x + y * y + z * z) <= 4.0)
    # include mode zeroes what projects off screen; q3 <= 0 (behind the camera) is left alone
    out = np.full((2, 3, 4), 255, np.uint8)
    m2 = m.copy()
    m2[0, 3] = 5.0
    oracle.mask_cut(out, 1.0, 1.0, 1.0, 1e9, np.zeros((8, 10), bool), m2, mv, 0)
    assert not out.any()
    out = np.full((2, 3, 4), 255, np.uint8)
    m3 = m.copy()
    m3[3, 3] = -1.0
    oracle.mask_cut(out, 1.0, 1.0, 1.0, 1e9, np.ones((8, 10), bool), m3, mv, 0)
    assert (out == 255).all()


def test_count_regions(oracle):
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 7, (5, 6, 7))
    got = oracle.count_regions(lab, 6)
    want = np.bincount(lab.ravel(), minlength=7)[lab]
    assert got.dtype == np.uint32 and np.array_equal(got, want)
    with pytest.raises(IndexError):
        oracle.count_regions(lab, 5)
    with pytest.raises(IndexError):
        oracle.count_regions(lab - 1, 6)


def test_convolve_non_zero_and_area_anchors(oracle):
    from scipy import ndimage
    rng = np.random.default_rng(4)
    vol = rng.normal(size=(7, 8, 9)) * (rng.random((7, 8, 9)) < 0.6)
    ker = rng.normal(size=(3, 3, 3))
    got = oracle.convolve_non_zero(vol, ker, -3)
    ref = np.where(vol != 0, ndimage.correlate(vol, ker, mode="constant", cval=-3.0), 0.0)
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-12)
    # even-sized kernel: centre offset is size // 2, like scipy's origin 0
    ker2 = rng.normal(size=(2, 4, 3))
    got2 = oracle.convolve_non_zero(vol, ker2, 0)
    ref2 = np.where(vol != 0, ndimage.correlate(vol, ker2, mode="constant", cval=0.0), 0.0)
    assert np.allclose(got2, ref2, rtol=1e-12, atol=1e-12)
    # area of an a x b x c voxel box: its exposed faces (cval = 1: the volume border does not count as exposed)
    m = np.zeros((12, 13, 14), np.uint8)
    m[3:6, 4:9, 5:12] = 255  # 3 (z) x 5 (y) x 7 (x) voxels
    sx, sy, sz = 0.5, 0.75, 2.0
    area = oracle.calc_image_area(m, (sx, sy, sz))
    nz, ny, nx = 3, 5, 7   # mask_matrix[1:,1:,1:] shifts the box, not its size
    want = 2 * (nx * ny) * (sx * sy) + 2 * (nx * nz) * (sx * sz) + 2 * (ny * nz) * (sy * sz)
    assert area == pytest.approx(want, rel=1e-12)
