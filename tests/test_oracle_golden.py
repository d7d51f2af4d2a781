"""Pin the CPU oracle against every golden vector the reference's own tests hold for the hot path
(SURVEY.md section 8c).  CPU only.  Reference test files cited per test."""
import numpy as np
from scipy import ndimage
from scipy.ndimage import generate_binary_structure

BONE = (226, 3071)  # invesalius/presets.py:37, tests/test_bone_thresholding.py:42-49


# ---- tests/test_segmentation_tools.py:17-51 ---------------------------------------------------
def test_region_growing_threshold(oracle):
    image = np.array([[[1, 1, 1, 5, 5], [1, 2, 2, 5, 5], [1, 2, 3, 5, 5], [1, 2, 2, 5, 5], [1, 1, 1, 5, 5]]],
                     dtype=np.int16)
    out_mask = np.zeros((1, 5, 5), dtype=np.uint8)
    oracle.floodfill_threshold(image, [[2, 2, 0]], 2, 3, 1, generate_binary_structure(3, 1), out_mask)
    expected = np.array([[0, 0, 0, 0, 0], [0, 1, 1, 0, 0], [0, 1, 1, 0, 0], [0, 1, 1, 0, 0], [0, 0, 0, 0, 0]],
                        dtype=np.uint8)
    assert np.array_equal(out_mask[0], expected)


# ---- tests/test_segmentation_tools.py:54-102 --------------------------------------------------
def test_region_growing_strct_disconnected(oracle):
    image = np.array([[[2, 2, 0], [0, 2, 0], [0, 0, 2]]], dtype=np.int16)
    seed = [[0, 0, 0]]
    out8 = np.zeros((1, 3, 3), dtype=np.uint8)
    oracle.floodfill_threshold(image, seed, 2, 2, 1, generate_binary_structure(3, 2), out8)
    assert np.array_equal(out8, np.array([[[1, 1, 0], [0, 1, 0], [0, 0, 1]]], dtype=np.uint8))
    out4 = np.zeros((1, 3, 3), dtype=np.uint8)
    oracle.floodfill_threshold(image, seed, 2, 2, 1, generate_binary_structure(3, 1), out4)
    assert np.array_equal(out4, np.array([[[1, 1, 0], [0, 1, 0], [0, 0, 0]]], dtype=np.uint8))


# ---- tests/test_segmentation_tools.py:105-134 -------------------------------------------------
def test_fill_holes_automatically(oracle):
    mask_2d = np.ones((7, 7), dtype=np.uint8)
    mask_2d[3, 3] = 0
    mask = mask_2d[np.newaxis, ...]
    labels_2d, nlabels = ndimage.label(mask_2d == 0, structure=np.ones((3, 3), dtype=np.uint8), output=np.uint32)
    border = set()
    for i in range(7):
        border.update([labels_2d[i, 0], labels_2d[i, -1], labels_2d[0, i], labels_2d[-1, i]])
    for bl in border:
        labels_2d[labels_2d == bl] = 0
    labels = labels_2d[np.newaxis, ...]
    ret = oracle.fill_holes_automatically(mask, labels, int(labels.max()), 1)
    expected = np.ones((1, 7, 7), dtype=np.uint8)
    expected[0, 3, 3] = 254
    assert ret
    assert np.array_equal(mask, expected)


# ---- tests/test_bone_thresholding.py:51-89,156-185 (per-slice preview incl. inclusive bounds) -
def test_set_mask_threshold_slice(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, BONE[0] - 1, (10, 10), dtype=np.int16)
    img[5:8, 5:8] = (BONE[0] + BONE[1]) // 2
    exp = np.zeros((10, 10), np.uint8)
    exp[5:8, 5:8] = 255
    assert np.array_equal(oracle.set_mask_threshold_slice(img, BONE), exp)
    edge = np.zeros((10, 10), dtype=np.int16)
    edge[0, :4] = [226, 3071, 225, 3072]
    got = oracle.set_mask_threshold_slice(edge, BONE)
    assert list(got[0, :4]) == [255, 255, 0, 0] and got[1:].sum() == 0


# ---- tests/test_bone_thresholding.py:92-118 ---------------------------------------------------
def test_do_threshold_to_a_slice(oracle):
    rng = np.random.default_rng(1)
    sl = rng.integers(0, BONE[0] - 1, (10, 10), dtype=np.int16)
    sl[5:8, 5:8] = (BONE[0] + BONE[1]) // 2
    m = np.zeros((10, 10), np.uint8)
    m[0:2, 0:2] = 1
    m[2:4, 2:4] = 2
    m[4:6, 4:6] = 253
    m[6:8, 6:8] = 254
    exp = np.zeros((10, 10), np.uint8)
    exp[5:8, 5:8] = 255
    exp[0:2, 0:2] = 1
    exp[2:4, 2:4] = 2
    exp[4:6, 4:6] = 253
    exp[6:8, 6:8] = 254
    assert np.array_equal(oracle.do_threshold_to_a_slice(sl, m, BONE), exp)


# ---- tests/test_bone_thresholding.py:121-153, tests/test_segmentation_tools.py:137-167 --------
def test_do_threshold_to_all_slices(oracle):
    rng = np.random.default_rng(2)
    vol = rng.integers(0, BONE[0] - 1, (10, 10, 10), dtype=np.int16)
    vol[5:8, 5:8, 5:8] = (BONE[0] + BONE[1]) // 2
    mask = np.zeros((11, 11, 11), np.uint8)  # tests/test_mask.py: shape + 1 per axis
    oracle.do_threshold_to_all_slices(mask, vol, BONE)
    exp = np.zeros((10, 10, 10), np.uint8)
    exp[5:8, 5:8, 5:8] = 255
    assert np.array_equal(mask[1:, 1:, 1:], exp)
    assert np.all(mask[1:, 0, 0] == 1)
    image = np.zeros((5, 5, 5), dtype=np.int16)
    image[2, 2, 2] = 100
    image[3, 3, 3] = 200
    mask = np.zeros((6, 6, 6), np.uint8)
    oracle.do_threshold_to_all_slices(mask, image, (100, 200))
    exp = np.zeros((5, 5, 5), np.uint8)
    exp[2, 2, 2] = exp[3, 3, 3] = 255
    assert np.array_equal(mask[1:, 1:, 1:], exp)


# ---- LUT: values measured with numpy (W=400,L=300) ---------------------------------
def test_lut_values(oracle):
    d = np.array([299, 301, 400, 3000, -1000], dtype=np.int16)
    assert list(oracle.get_LUT_value_255(d, 400, 300)) == [127, 128, 191, 255, 0]
    assert oracle.get_LUT_value(d, 400, 300).dtype == np.int16


# ---- tests/test_mesh_generation.py:23-37: 20^3 cube [5:15]^3 @ iso 128 -> closed surface -------
def _closed_and_oriented(tris):
    """every directed edge must appear exactly once in each direction (watertight, consistently oriented)"""
    v = tris.reshape(-1, 3)
    uniq, inv = np.unique(v, axis=0, return_inverse=True)
    f = inv.reshape(-1, 3)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    e = e[e[:, 0] != e[:, 1]]  # drop collapsed edges of degenerate triangles
    fwd = {}
    for a, b in e:
        fwd[(a, b)] = fwd.get((a, b), 0) + 1
    return all(fwd.get((b, a), 0) == n for (a, b), n in fwd.items())


def _signed_volume(tris):
    t = tris.astype(np.float64)
    return np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0


def test_marching_cubes_cube(oracle):
    m = np.zeros((20, 20, 20), np.uint8)
    m[5:15, 5:15, 5:15] = 255
    tris = oracle.marching_cubes(m, (1.0, 1.0, 1.0), [128.0], 0, False, False, False, 0.0, 0)
    assert len(tris) > 0
    assert _closed_and_oriented(tris)
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    t = 128.0 / 255.0
    # x and z un-flipped: 4+t .. 15-t ; y flipped about the origin (tests/test_mesh_generation.py:77-86)
    np.testing.assert_allclose([lo[0], hi[0], lo[2], hi[2]], [4 + t, 15 - t, 4 + t, 15 - t], rtol=0, atol=1e-6)
    np.testing.assert_allclose([lo[1], hi[1]], [-(15 - t), -(4 + t)], rtol=0, atol=1e-6)
    # outward orientation (normals from >=iso to <iso): positive enclosed volume, ~ (10+2(0.5-t)...)^3
    vol = _signed_volume(tris)
    assert 9.0 ** 3 < vol < 11.0 ** 3


def test_marching_cubes_random_watertight(oracle):
    """all 256 cases: a random field padded with the minimum must give a closed, oriented surface"""
    rng = np.random.default_rng(3)
    a = rng.integers(-1000, 1000, (9, 10, 11)).astype(np.int16)
    tris = oracle.marching_cubes(a, (0.5, 0.75, 2.0), [0.5], 0, True, True, True, float(np.iinfo(np.int16).min), 1)
    assert len(tris) > 1000
    assert _closed_and_oriented(tris)
    assert _signed_volume(tris) > 0


def test_marching_cubes_pieces_concatenate(oracle):
    """surface.py:1362-1380 piece split (20 slices + 1 overlap): pieces tile the whole-volume soup"""
    rng = np.random.default_rng(4)
    img = (rng.normal(0, 300, (45, 12, 13))).astype(np.int16)
    mask = np.zeros((46, 13, 14), np.uint8)
    mask[1:, 1:, 1:] = np.where(img > 100, 255, 0)
    whole = oracle.create_surface_piece(img, mask, slice(0, 45), (1, 1, 2), 0, 0, True)
    parts = []
    n_pieces = int(round(45 / 20 + 0.5, 0))
    for i in range(n_pieces):
        roi = slice(i * 20, i * 20 + 21)
        parts.append(oracle.create_surface_piece(img, mask, roi, (1, 1, 2), 0, 0, True))
    cat = np.concatenate(parts)
    key = lambda t: np.sort(t.reshape(len(t), -1).view([("", np.float32)] * 9), axis=0)
    assert len(cat) == len(whole)
    assert np.array_equal(key(cat), key(whole))


# ---- apply_view_matrix_transform: no reference test exists; cross-check the restatement against scipy ---------------
def test_view_transform_trilinear_matches_scipy_map_coordinates(oracle):
    """independent check of the oracle (transforms.rs / interpolation.rs restatement): for sample points strictly inside
    the volume, trilinear resampling == scipy.ndimage.map_coordinates(order=1) up to float rounding"""
    rng = np.random.default_rng(21)
    vol = rng.normal(0, 100, (14, 16, 18))
    th = 0.4
    R = np.eye(4)
    R[1:3, 1:3] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    c = np.array([7.0, 8.0, 9.0])
    T0, T1 = np.eye(4), np.eye(4)
    T0[:3, 3], T1[:3, 3] = -c, c
    M = np.ascontiguousarray(T1 @ R @ T0)
    out = np.zeros((4, 16, 18))
    oracle.apply_view_matrix_transform(vol, (1.0, 1.0, 1.0), M, 5, "AXIAL", 1, -1e9, out)
    zz, yy, xx = np.meshgrid(np.arange(5, 9), np.arange(16), np.arange(18), indexing="ij")
    pts = np.stack([zz, yy, xx, np.ones_like(zz)]).reshape(4, -1).astype(float)
    q = M @ pts
    ref = ndimage.map_coordinates(vol, q[:3], order=1, mode="nearest").reshape(out.shape)
    inside = (out != -1e9)
    assert inside.mean() > 0.5
    np.testing.assert_allclose(out[inside], ref[inside], rtol=0, atol=1e-9)
    # nearest = truncation of the transformed coordinate
    outn = np.zeros((4, 16, 18))
    oracle.apply_view_matrix_transform(vol, (1.0, 1.0, 1.0), M, 5, "AXIAL", 0, -1e9, outn)
    qi = q[:3].astype(int).reshape(3, *out.shape)
    assert np.array_equal(outn[inside], vol[qi[0][inside], qi[1][inside], qi[2][inside]])


def test_floodfill_equal_value_and_auto_threshold_hand_cases(oracle):
    """floodfill.rs:5-49 and floodfill_py.rs:12-85 have no test in the reference tree (parity unpinned: the restatement
    is checked against cases worked out by hand from the Rust source)."""
    # floodfill: component of data == v; the seed is filled whatever its value; out == fill is a barrier
    d = np.array([[[5, 7, 7, 7, 9, 7]]], np.int16)
    o = np.zeros(d.shape, np.uint8)
    oracle.floodfill(d, 0, 0, 0, 7, 3, o)            # seed holds 5, not 7: still filled and expanded
    assert o.tolist() == [[[3, 3, 3, 3, 0, 0]]]
    o = np.zeros(d.shape, np.uint8)
    o[0, 0, 2] = 3                                   # pre-filled voxel stops the walk
    oracle.floodfill(d, 1, 0, 0, 7, 3, o)
    assert o.tolist() == [[[0, 3, 3, 0, 0, 0]]]
    # auto threshold, p = 0.05: a step leaves a voxel of value v into [ceil(0.95 v), floor(1.05 v)]
    d = np.array([[[100, 104, 109, 120, 121]]], np.int16)
    o = np.zeros(d.shape, np.uint8)
    oracle.floodfill_auto_threshold(d, [(0, 0, 0)], 0.05, 1, o)
    assert o.tolist() == [[[1, 1, 1, 0, 0]]]         # 100 -> 104 -> 109, but 109 -> 120 is out of [104, 114]
    # the relation is directed: 200 -> 190 is allowed ([190, 210]), 190 -> 200 is not ([181, 199])
    d = np.array([[[200, 190]]], np.int16)
    o = np.zeros(d.shape, np.uint8)
    oracle.floodfill_auto_threshold(d, [(0, 0, 0)], 0.05, 1, o)
    assert o.tolist() == [[[1, 1]]]
    o = np.zeros(d.shape, np.uint8)
    oracle.floodfill_auto_threshold(d, [(1, 0, 0)], 0.05, 1, o)
    assert o.tolist() == [[[0, 1]]]
    # negative values flip the products: v = -100 -> [ceil(-95), floor(-105)] = [-95, -105] is empty
    d = np.array([[[-100, -100]]], np.int16)
    o = np.zeros(d.shape, np.uint8)
    oracle.floodfill_auto_threshold(d, [(0, 0, 0)], 0.05, 1, o)
    assert o.tolist() == [[[1, 0]]]
    # `as i16` saturates: 30000 * 1.5 = 45000 -> 32767
    d = np.array([[[30000, 32767, 14000]]], np.int16)
    o = np.zeros(d.shape, np.uint8)
    oracle.floodfill_auto_threshold(d, [(0, 0, 0)], 0.5, 1, o)
    assert o.tolist() == [[[1, 1, 0]]]               # 32767 -> 14000 is below ceil(16383.5)


def test_jump_flooding_is_the_voronoi_diagram_on_a_small_grid(oracle):
    """floodfill.rs:298-507 has no test in the reference tree (parity unpinned); on an 8^3 grid with two sites the passes
    reach every voxel, so the restatement must reproduce the brute-force Voronoi diagram and distances."""
    shape = (8, 8, 8)
    d = np.full(shape, -1, np.float32)
    o = np.zeros(shape, np.int32)
    sites = np.array([[1, 1, 1], [6, 6, 6]], np.int32)
    oracle.jump_flooding(d, o, sites, False)
    zz, yy, xx = np.meshgrid(*[np.arange(8)] * 3, indexing="ij")
    d1 = np.sqrt(((zz - 1) ** 2 + (yy - 1) ** 2 + (xx - 1) ** 2).astype(np.float32))
    d2 = np.sqrt(((zz - 6) ** 2 + (yy - 6) ** 2 + (xx - 6) ** 2).astype(np.float32))
    assert np.array_equal(o, np.where(d1 <= d2, 1, 2))
    assert np.array_equal(d, np.minimum(d1, d2).astype(np.float32))
    oracle.jump_flooding(d, o, sites, True)          # normalised: 1.0 at the farthest voxel of each cell
    assert abs(float(d[o == 1].max()) - 1.0) < 1e-6 and abs(float(d[o == 2].max()) - 1.0) < 1e-6
