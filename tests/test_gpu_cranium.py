"""BASELINE configs[0] in shape: the reference's sample case (Cranium.inv3: 108 x 512 x 512 int16, spacing
(0.4785156, 0.4785156, 2.0), bone threshold 226..3071, surface -> STL) is not in the mount, so a cranium-like phantom of
the same geometry stands in.  The whole chain runs through the host-level entry points the GUI would
call and is compared with the CPU oracle; the assertions of the reference's own tests (test_bone_thresholding.py,
test_mesh_generation.py, test_stl_export.py: mask values, non-empty surface, bounds, STL round trip) are restated."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPACING = (0.4785156, 0.4785156, 2.0)
BONE = (226, 3071)


def cranium_phantom(shape=(108, 512, 512), seed=7):
    """air -1000, soft tissue 40 +- 20 inside an ellipsoidal head, a bone shell of 700..1800 HU, two 'orbits' cut out"""
    rng = np.random.default_rng(seed)
    dz, dy, dx = shape
    z, y, x = np.ogrid[:dz, :dy, :dx]
    r = np.sqrt(((z - dz * 0.5) / (dz * 0.62)) ** 2 + ((y - dy * 0.5) / (dy * 0.40)) ** 2 + ((x - dx * 0.5) / (dx * 0.33)) ** 2)
    img = np.full(shape, -1000, np.int16)
    head = r < 1.0
    img[head] = (40 + rng.normal(0, 20, int(head.sum()))).astype(np.int16)
    shell = (r > 0.90) & (r < 1.0)
    img[shell] = rng.integers(700, 1800, int(shell.sum())).astype(np.int16)
    for cx in (0.36, 0.64):
        orbit = ((y - dy * 0.30) ** 2 + (x - dx * cx) ** 2 < (dx * 0.06) ** 2) & (z > dz * 0.3) & (z < dz * 0.6)
        img[np.broadcast_to(orbit, shape) & shell] = 40
    return img


def test_bone_mask_surface_and_stl(ivxlib, oracle, tmp_path):
    from invesalius3_amd import slice_ as sl
    from invesalius3_amd import surface_process as sp
    img = cranium_phantom()
    # -- bone threshold into the flag-carrying mask matrix (do_threshold_to_all_slices) ---------------------------------
    mask = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    sl.do_threshold_to_all_slices(mask, img, BONE)
    want = np.zeros_like(mask)
    oracle.do_threshold_to_all_slices(want, img, BONE)
    assert np.array_equal(mask, want)
    inner = mask[1:, 1:, 1:]
    assert set(np.unique(inner)) == {0, 255}
    assert np.array_equal(inner == 255, (img >= 226) & (img <= 3071))  # inclusive bounds: 225 / 226 / 3071 / 3072
    assert (mask[1:, 0, 0] == 1).all() and 0.01 < (inner == 255).mean() < 0.2
    # -- surface in the reference's 20+1-slice pieces --------------------------------------------------------------------
    soup = sp.create_surface(None, mask, SPACING, 0, 0, True)
    ref = np.concatenate([oracle.create_surface_piece(None, mask, slice(i * 20, i * 20 + 21), SPACING, 0, 0, True)
                          for i in range(int(round(img.shape[0] / 20 + 0.5, 0))) if i * 20 < img.shape[0]])
    assert soup.shape == ref.shape and np.array_equal(soup, ref)
    assert len(soup) > 100000
    # bounds as test_mesh_generation.py checks them: x in [0, w*sx], y flipped (<= 0), z in [0, d*sz] (+- the padding)
    lo, hi = soup.reshape(-1, 3).min(0), soup.reshape(-1, 3).max(0)
    assert -SPACING[0] <= lo[0] and hi[0] <= img.shape[2] * SPACING[0]
    assert hi[1] <= SPACING[1] and lo[1] >= -img.shape[1] * SPACING[1]
    assert -SPACING[2] <= lo[2] and hi[2] <= img.shape[0] * SPACING[2] + SPACING[2]
    # -- merged surface, largest region, measurements ----------------------------------------------------------------------
    verts, faces, meas = sp.join_process_volume(None, mask, SPACING, 0, 0, True, keep_largest_region=True)
    v0, f0 = sp.marching_cubes_indexed(mask[1:, 1:, 1:], SPACING, [127.0])
    vk, fk, nreg = oracle.mesh_keep_largest(v0, f0)
    assert np.array_equal(verts, vk) and np.array_equal(faces, fk)
    m = oracle.mesh_mass_properties(vk, fk)
    assert meas["volume"] == pytest.approx(m[0], rel=1e-10) and meas["area"] == pytest.approx(m[1], rel=1e-10)
    # the shell is 10 % of the semi-axes thick: its volume is (1 - 0.9^3) of the ellipsoid's, give or take the orbits
    a, b, c = 108 * 0.62 * 2.0, 512 * 0.40 * 0.4785156, 512 * 0.33 * 0.4785156
    shell_vol = 4 / 3 * np.pi * a * b * c * (1 - 0.9 ** 3)
    clipped = 0.5  # the head is taller than the 108-slice stack: roughly the middle half of the ellipsoid is present
    assert 0.3 * clipped * shell_vol < meas["volume"] < 1.5 * shell_vol
    # -- STL export round trip (test_stl_export.py): 80-byte header, count, 50 bytes per triangle ------------------------
    path = tmp_path / "bone.stl"
    sp.write_stl_binary(path, soup)
    raw = path.read_bytes()
    n = struct.unpack("<I", raw[80:84])[0]
    assert n == len(soup) and len(raw) == 84 + 50 * n
    rec = np.frombuffer(raw[84:], dtype=[("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    assert np.array_equal(rec["v"], soup)
    nn = np.linalg.norm(rec["n"], axis=1)
    assert np.all((np.abs(nn - 1) < 1e-4) | (nn == 0))
