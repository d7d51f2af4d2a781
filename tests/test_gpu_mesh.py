"""Surface post-processing on the indexed mesh: keep-largest region and mass properties (join_process_surface,
invesalius/data/surface_process.py:376-391, 452-458) -- HIP path vs the CPU oracle, plus analytic anchors."""
import numpy as np
import pytest

from conftest import synth_volume

pytestmark = pytest.mark.gpu


def _ball_mask(n, centres, radii):
    z, y, x = np.mgrid[:n, :n, :n]
    m = np.zeros((n, n, n), np.uint8)
    for (cz, cy, cx), r in zip(centres, radii):
        m[(z - cz) ** 2 + (y - cy) ** 2 + (x - cx) ** 2 <= r * r] = 255
    return m


def test_keep_largest_matches_oracle_on_blobs(ivxlib, oracle):
    from invesalius3_amd import surface_process as sp
    m = _ball_mask(48, [(12, 12, 12), (30, 30, 30), (10, 36, 36)], [6, 11, 4])
    verts, faces = sp.marching_cubes_indexed(m, (1.0, 1.0, 1.0), [127.0])
    v1, f1, nreg = sp.keep_largest(verts, faces)
    v0, f0, nreg0 = oracle.mesh_keep_largest(verts, faces)
    assert nreg == nreg0 == 3
    assert np.array_equal(f1, f0) and np.array_equal(v1, v0)
    # it is the big ball: its triangle soup is a subsequence of the full soup
    assert 0 < len(f1) < len(faces)
    vol, area = sp.mass_properties(v1, f1)
    assert abs(vol - 4 / 3 * np.pi * 11 ** 3) / (4 / 3 * np.pi * 11 ** 3) < 0.03


def test_keep_largest_noisy_surface_and_tie_break(ivxlib, oracle):
    from invesalius3_amd import surface_process as sp
    img = synth_volume((40, 48, 80), seed=9)
    img[np.random.default_rng(3).random(img.shape) < 0.003] = 2500  # specks: many small regions
    verts, faces = sp.marching_cubes_indexed(img, (0.5, 0.5, 1.0), [300.0], 0, True, True, True,
                                             float(np.iinfo(np.int16).min), 1)
    v1, f1, nreg = sp.keep_largest(verts, faces)
    v0, f0, nreg0 = oracle.mesh_keep_largest(verts, faces)
    assert nreg == nreg0 and nreg > 1
    assert np.array_equal(f1, f0) and np.array_equal(v1, v0)
    # two identical cubes: equally many triangles -> the first one in triangle order is kept
    m = np.zeros((12, 12, 40), np.uint8)
    m[3:8, 3:8, 5:10] = 255
    m[3:8, 3:8, 25:30] = 255
    verts, faces = sp.marching_cubes_indexed(m, (1.0, 1.0, 1.0), [127.0])
    v1, f1, nreg = sp.keep_largest(verts, faces)
    v0, f0, _ = oracle.mesh_keep_largest(verts, faces)
    assert nreg == 2 and len(f1) == len(faces) // 2
    assert np.array_equal(f1, f0) and np.array_equal(v1, v0)
    assert np.array_equal(v1[f1][0], verts[faces][0])


def test_keep_largest_degenerate_inputs(ivxlib):
    from invesalius3_amd import surface_process as sp
    v, f, n = sp.keep_largest(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
    assert v.shape == (0, 3) and f.shape == (0, 3) and n == 0
    # unused vertices are dropped; a single triangle is its own region
    verts = np.arange(15, dtype=np.float32).reshape(5, 3)
    v, f, n = sp.keep_largest(verts, np.array([[4, 1, 3]], np.int32))
    assert n == 1 and np.array_equal(f, [[2, 0, 1]]) and np.array_equal(v, verts[[1, 3, 4]])
    with pytest.raises((ValueError, IndexError)):
        sp.keep_largest(verts, np.array([[0, 1, 5]], np.int32))


@pytest.mark.parametrize("indexed", [True, False])
def test_mass_properties_match_oracle(ivxlib, oracle, indexed):
    from invesalius3_amd import surface_process as sp
    img = synth_volume((33, 40, 72), seed=4)
    args = ((0.4785156, 0.4785156, 2.0), [250.0], 7, True, True, True, float(np.iinfo(np.int16).min), 1)
    verts, faces = sp.marching_cubes_indexed(img, *args)
    if indexed:
        got = sp.mass_properties_full(verts, faces)
        want = oracle.mesh_mass_properties(verts, faces)
    else:
        soup = verts[faces]
        got = sp.mass_properties_full(soup)
        want = oracle.mesh_mass_properties(soup)
    # double sums in a different (tree) order: 1e-10 relative; the integer weights are exact
    assert got[5:] == pytest.approx(want[5:], abs=0, rel=1e-15)
    for g, w in zip(got[:5], want[:5]):
        assert g == pytest.approx(w, rel=1e-10, abs=1e-9)
    # closed surface: each axis' divergence sum is the same volume, and equals the signed-tetrahedra volume
    t = verts[faces].astype(np.float64)
    tet = abs(np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0)
    assert got[0] == pytest.approx(tet, rel=1e-9)
    assert abs(got[2]) == pytest.approx(tet, rel=1e-9) and abs(got[4]) == pytest.approx(tet, rel=1e-9)


def test_mass_properties_cube_is_exact(ivxlib):
    from invesalius3_amd import surface_process as sp
    # a 5x4x3-voxel box mask: the iso-127 surface of a binary mask cuts every crossing edge at t = 127/255
    m = np.zeros((9, 10, 11), np.uint8)
    m[3:6, 3:7, 3:8] = 255
    verts, faces = sp.marching_cubes_indexed(m, (1.0, 1.0, 1.0), [127.0])
    vol, area = sp.mass_properties(verts, faces)
    soup_vol, soup_area = sp.mass_properties(verts[faces])
    assert vol == pytest.approx(soup_vol, rel=1e-12) and area == pytest.approx(soup_area, rel=1e-12)
    # between the inner box (4x3x2 cell centres) and the outer one
    assert 4 * 3 * 2 < vol < 6 * 5 * 4
    assert sp.mass_properties(np.zeros((0, 3, 3), np.float32)) == (0.0, 0.0)


def test_device_pipeline_indexed_then_largest_then_mass(ivxlib, oracle):
    """device-resident chain on one stream: mask -> indexed mesh -> largest region -> area/volume"""
    import ctypes
    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceBuffer, DeviceVolume
    img = synth_volume((48, 64, 64), seed=12)
    vol = DeviceVolume(img, spacing=(1.0, 1.0, 1.0))
    vol.threshold(200, 3071)
    nv, nt = vol.marching_cubes_indexed()
    ov, of, out = DeviceBuffer(nv * 12 + 16), DeviceBuffer(nt * 12 + 16), DeviceBuffer(64)
    n1, n2, nr = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    lib = L.lib()
    L.check(lib.ivx_dev_mesh_keep_largest(vol._verts.ptr, ctypes.c_int64(nv), vol._faces.ptr, ctypes.c_int64(nt), ov.ptr,
                                          ctypes.c_int64(nv), of.ptr, ctypes.c_int64(nt), ctypes.byref(n1),
                                          ctypes.byref(n2), ctypes.byref(nr), vol.stream))
    L.check(lib.ivx_dev_mesh_mass_properties(ov.ptr, of.ptr, n2, out.ptr, vol.stream))
    vol.sync()
    verts = vol._verts.download((nv, 3), np.float32)
    faces = vol._faces.download((nt, 3), np.int32)
    v0, f0, nr0 = oracle.mesh_keep_largest(verts, faces)
    assert (n1.value, n2.value, nr.value) == (len(v0), len(f0), nr0)
    assert np.array_equal(ov.download((n1.value, 3), np.float32), v0)
    assert np.array_equal(of.download((n2.value, 3), np.int32), f0)
    want = oracle.mesh_mass_properties(v0, f0)
    got = out.download((8,), np.float64)
    assert got[0] == pytest.approx(want[0], rel=1e-10) and got[1] == pytest.approx(want[1], rel=1e-10)
    vol.close()


def test_join_process_volume_equals_the_appended_pieces(ivxlib, oracle):
    """pieces of 20 slices + 1 overlap, appended (create_surface) == one pass over the volume, merged"""
    from invesalius3_amd import surface_process as sp
    img = synth_volume((45, 40, 48), seed=23)
    mask = np.zeros((46, 41, 49), np.uint8)
    mask[1:, 1:, 1:] = np.where(img > 150, 255, 0)
    soup = sp.create_surface(None, mask, (0.5, 0.5, 1.5), 0, 0, True)
    verts, faces, m = sp.join_process_volume(None, mask, (0.5, 0.5, 1.5), 0, 0, True)
    assert np.array_equal(verts[faces], soup)
    want = oracle.mesh_mass_properties(soup)
    assert m["volume"] == pytest.approx(want[0], rel=1e-9) and m["area"] == pytest.approx(want[1], rel=1e-9)
    v2, f2, m2 = sp.join_process_volume(None, mask, (0.5, 0.5, 1.5), 0, 0, True, keep_largest_region=True)
    assert len(f2) <= len(faces) and m2["area"] <= m["area"] + 1e-9
    # image path: two iso-values
    soup = sp.create_surface(img, mask, (1.0, 1.0, 1.0), 100, 900, False)
    verts, faces, _ = sp.join_process_volume(img, mask, (1.0, 1.0, 1.0), 100, 900, False)
    # each piece emits iso 0 then iso 1, the whole volume all of iso 0 then all of iso 1: same triangles, other order
    key = lambda t: t[np.lexsort(t.reshape(len(t), 9).T[::-1])]
    assert np.array_equal(key(verts[faces]), key(soup))
