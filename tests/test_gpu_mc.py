"""GPU parity: marching cubes vs the C oracle (same case table, same double->float32 rounding): the soup must be
IDENTICAL, triangle by triangle (stronger than the 1e-5 vertex tolerance of the north star)."""
import numpy as np
import pytest

from conftest import synth_volume

pytestmark = pytest.mark.gpu


def _cmp(gpu, ref):
    assert gpu.shape == ref.shape
    assert np.array_equal(gpu, ref)


@pytest.mark.parametrize("shape,pads", [((9, 10, 11), (True, True, True)), ((4, 5, 70), (True, False, True)),
                                        ((3, 3, 3), (False, False, False)), ((6, 64, 64), (True, True, False)),
                                        ((2, 2, 130), (False, False, False)), ((21, 48, 56), (True, True, True))])
def test_random_field_all_cases(ivxlib, oracle, shape, pads):
    from invesalius3_amd import surface_process as sp
    rng = np.random.default_rng(3)
    a = rng.integers(-1000, 1000, shape).astype(np.int16)
    pxy, pb, pt = pads
    args = ((0.5, 0.75, 2.0), [0.5], 7, pxy, pb, pt, float(np.iinfo(np.int16).min), int(pxy and pb))
    _cmp(sp.marching_cubes(a, *args), oracle.marching_cubes(a, *args))


def test_two_iso_default_mode(ivxlib, oracle):
    from invesalius3_amd import surface_process as sp
    img = synth_volume((24, 40, 48), seed=21)
    mask = np.zeros((25, 41, 49), np.uint8)
    for roi in (slice(0, 21), slice(20, 41)):
        g = sp.surface_piece(img, mask, roi, (0.4785156, 0.4785156, 2.0), 226, 3071, False)
        r = oracle.create_surface_piece(img, mask, roi, (0.4785156, 0.4785156, 2.0), 226, 3071, False)
        _cmp(g, r)
        assert len(g) > 0


def test_binary_mask_pieces_and_reference_fixture(ivxlib, oracle):
    """tests/test_mesh_generation.py:23-37 (20^3 cube @ iso 128 -> closed surface, bounds) + piece split."""
    from invesalius3_amd import surface_process as sp
    m = np.zeros((20, 20, 20), np.uint8)
    m[5:15, 5:15, 5:15] = 255
    t = sp.marching_cubes(m, (1.0, 1.0, 1.0), [128.0], 0, False, False, False, 0.0, 0)
    _cmp(t, oracle.marching_cubes(m, (1.0, 1.0, 1.0), [128.0], 0, False, False, False, 0.0, 0))
    lo, hi = t.reshape(-1, 3).min(0), t.reshape(-1, 3).max(0)
    f = 128.0 / 255.0
    np.testing.assert_allclose([lo[0], hi[0], lo[2], hi[2]], [4 + f, 15 - f, 4 + f, 15 - f], atol=1e-6)
    np.testing.assert_allclose([lo[1], hi[1]], [-(15 - f), -(4 + f)], atol=1e-6)

    img = synth_volume((45, 32, 40), seed=22)
    mask = np.zeros((46, 33, 41), np.uint8)
    mask[1:, 1:, 1:] = np.where(img > -700, 255, 0)
    g = sp.create_surface(None, mask, (1, 1, 2), 0, 0, True)
    parts = [oracle.create_surface_piece(None, mask, slice(i * 20, i * 20 + 21), (1, 1, 2), 0, 0, True)
             for i in range(3)]
    _cmp(g, np.concatenate(parts))


def test_strided_mask_view_and_empty(ivxlib, oracle):
    from invesalius3_amd import surface_process as sp
    mask = np.zeros((11, 21, 31), np.uint8)
    mask[3:8, 4:15, 6:25] = 255
    a = mask[1:, 1:, 1:]  # the non-contiguous view the reference passes around
    _cmp(sp.marching_cubes(a, (1, 1, 1), [127.0]), oracle.marching_cubes(a, (1, 1, 1), [127.0]))
    z = np.zeros((5, 6, 7), np.uint8)
    assert len(sp.marching_cubes(z, (1, 1, 1), [127.0])) == 0
    assert len(sp.marching_cubes(np.zeros((0, 6, 7), np.uint8), (1, 1, 1), [127.0])) == len(
        oracle.marching_cubes(np.zeros((0, 6, 7), np.uint8), (1, 1, 1), [127.0]))


def test_full_size_properties_512(ivxlib):
    """512^3 binary mask of a ball: closed surface => signed volume ~ ball volume; vertex set invariant under
    the 20+1-slice piece decomposition (whole-volume soup == concatenated pieces as multisets)."""
    from invesalius3_amd import surface_process as sp
    n = 512
    z, y, x = np.ogrid[:n, :n, :n]
    ball = ((z - 250.5) ** 2 + (y - 260.25) ** 2 + (x - 240.75) ** 2) <= 180.0 ** 2
    mask = np.zeros((n + 1,) * 3, np.uint8)
    mask[1:, 1:, 1:] = np.where(ball, 255, 0)
    whole = sp.surface_piece(None, mask, slice(0, n), (1, 1, 1), 0, 0, True)
    t = whole.astype(np.float64)
    vol = np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0
    assert abs(vol / (4 / 3 * np.pi * 180.0 ** 3) - 1) < 5e-3
    pieces = sp.create_surface(None, mask, (1, 1, 1), 0, 0, True)
    assert len(pieces) == len(whole)
    key = lambda a: np.sort(a.reshape(len(a), -1).view([("", np.float32)] * 9), axis=0)
    assert np.array_equal(key(pieces), key(whole))


def test_empty_pieces_after_large_allocations(ivxlib, oracle):
    """regression: a piece that is empty along an axis is ALL padding; the branch-free corner loads must not touch
    memory outside the scratch block (this faulted the GPU when earlier calls had left larger workspaces behind)"""
    from invesalius3_amd import surface_process as sp
    big = np.zeros((40, 64, 64), np.int16)
    big[10:30, 10:50, 10:50] = 1000
    assert len(sp.marching_cubes(big, (1, 1, 1), [500.0, 900.0], pad_value=-32768.0)) > 0
    for shape in ((0, 6, 7), (5, 0, 7), (5, 6, 0), (0, 0, 0), (1, 1, 1)):
        for pads in ((True, True, True), (False, False, False), (True, False, True)):
            a = np.zeros(shape, np.uint8)
            g = sp.marching_cubes(a, (1, 1, 1), [127.0], 0, *pads, 255.0 if shape == (1, 1, 1) else 0.0)
            r = oracle.marching_cubes(a, (1, 1, 1), [127.0], 0, *pads, 255.0 if shape == (1, 1, 1) else 0.0)
            assert g.shape == r.shape and np.array_equal(g, r)


@pytest.mark.parametrize("case", ["random_i16", "mask_u8", "two_iso", "no_pad", "on_grid", "on_grid_pad"])
def test_indexed_mesh_is_the_merged_soup(ivxlib, oracle, case):
    """point merge (vtkCleanPolyData's job in join_process_surface): verts[faces] == the soup, bit for bit and in
    order; vertices are unique; their number equals the number of distinct soup vertices"""
    from invesalius3_amd import surface_process as sp
    rng = np.random.default_rng(17)
    if case == "random_i16":
        a = rng.integers(-1000, 1000, (9, 12, 70)).astype(np.int16)
        args = ((0.5, 0.75, 2.0), [0.5], 3, True, True, True, float(np.iinfo(np.int16).min), 1)
    elif case == "mask_u8":
        img = synth_volume((30, 40, 72), seed=81)
        a = np.where(img > -800, 255, 0).astype(np.uint8)
        a[rng.random(a.shape) < 0.01] = 254
        args = ((1.0, 1.0, 1.0), [127.0], 0, True, True, True, 0.0, 1)
    elif case == "two_iso":
        a = synth_volume((21, 40, 48), seed=82)
        args = ((0.4785156, 0.4785156, 2.0), [226.0, 3071.0], 20, True, False, True, float(np.iinfo(np.int16).min), 0)
    elif case == "on_grid":
        # many samples equal the iso-value: crossings land exactly on grid points and must share one vertex
        a = rng.integers(0, 5, (8, 9, 131)).astype(np.int16)
        args = ((1.0, 0.5, 0.25), [2.0, 3.0], 5, True, True, True, -5.0, 1)
    elif case == "on_grid_pad":
        # ... and so does the padding value
        a = rng.integers(0, 4, (5, 6, 64)).astype(np.uint8)
        args = ((1.0, 1.0, 1.0), [2.0], 0, True, True, True, 2.0, 1)
    else:
        a = rng.integers(0, 255, (6, 7, 130)).astype(np.uint8)
        args = ((1.0, 2.0, 3.0), [100.0], 0, False, False, False, 0.0, 0)
    verts, faces = sp.marching_cubes_indexed(a, *args)
    soup = oracle.marching_cubes(a, *args)
    assert faces.shape == (len(soup), 3) and faces.dtype == np.int32
    assert np.array_equal(verts[faces], soup)
    ni = len(args[1])
    if ni == 1:
        uniq = np.unique(soup.reshape(-1, 3), axis=0)
        assert len(verts) == len(uniq) == len(np.unique(verts, axis=0))
    else:
        # the two iso-surfaces are merged separately (their vertices never coincide geometrically unless a sample
        # sits between equal iso-values); check uniqueness inside each index range
        s0 = oracle.marching_cubes(a, args[0], args[1][:1], *args[2:])
        n0 = len(np.unique(s0.reshape(-1, 3), axis=0))
        s1 = oracle.marching_cubes(a, args[0], args[1][1:], *args[2:])
        n1 = len(np.unique(s1.reshape(-1, 3), axis=0))
        assert len(verts) == n0 + n1
        assert len(np.unique(verts[:n0], axis=0)) == n0 and len(np.unique(verts[n0:], axis=0)) == n1
    assert faces.min() == 0 and faces.max() == len(verts) - 1
    vol, area = sp.mass_properties(verts[faces])
    assert area > 0


@pytest.mark.parametrize("dtype", [np.uint8, np.int16])
def test_single_launch_surface_equals_count_list_emit_and_the_oracle(ivxlib, oracle, monkeypatch, dtype):
    """ivx_dev_mc_surface (k_mc_fused: count, look-back offsets and emit in one launch) against the four-launch path and the
    oracle, soup for soup: dense binary noise (thousands of triangles per workgroup -> several LDS windows, every output
    phase of the 16-byte stores), padded and unpadded pieces, rows that are not whole words, many workgroups (look-back
    across more than 64 predecessors)."""
    from invesalius3_amd import surface_process as sp
    rng = np.random.default_rng(11)
    for shape, pads, dens in [((12, 40, 200), (True, True, True), 0.5), ((5, 33, 130), (False, False, False), 0.5),
                              ((20, 64, 64), (True, False, True), 0.3), ((70, 96, 129), (True, True, True), 0.08),
                              ((3, 2, 2), (True, True, True), 0.5), ((40, 128, 192), (True, True, False), 0.5)]:
        if dtype == np.uint8:
            a = np.where(rng.random(shape) < dens, rng.integers(128, 256, shape), rng.integers(0, 127, shape)).astype(np.uint8)
            iso, padv = 127.0, 0.0
        else:
            a = np.where(rng.random(shape) < dens, rng.integers(300, 2000, shape), rng.integers(-1000, 299, shape)).astype(np.int16)
            iso, padv = 299.5, float(np.iinfo(np.int16).min)
        args = ((0.5, 0.75, 2.0), [iso], 7, *pads, padv, int(pads[0] and pads[1]))
        monkeypatch.setenv("IVX_MC_ONE_LAUNCH", "1")
        new = sp.marching_cubes(a, *args)
        monkeypatch.setenv("IVX_MC_ONE_LAUNCH", "0")
        old = sp.marching_cubes(a, *args)
        monkeypatch.delenv("IVX_MC_ONE_LAUNCH")
        assert len(new) > 0
        _cmp(new, old)
        _cmp(new, oracle.marching_cubes(a, *args))


def test_single_launch_surface_that_outgrows_its_buffer(ivxlib, oracle, monkeypatch):
    """A resident volume's triangle buffer comes from the previous call: a surface that outgrew it is written up to the
    capacity (nothing past it), the count says so, and the second launch into a larger buffer gives the whole soup."""
    from invesalius3_amd.device import DeviceVolume
    monkeypatch.setenv("IVX_MC_ONE_LAUNCH", "1")  # (opt-in path: measured slower than count + list + emit at 512^3)
    rng = np.random.default_rng(12)
    img = rng.integers(-1000, 1000, (40, 64, 128)).astype(np.int16)
    vol = DeviceVolume(img)
    for lo in (990, 0, 600, -500):  # sparse -> dense (outgrows) -> sparser (fits) -> densest
        vol.threshold(lo, 3071)
        got = vol.marching_cubes(from_binary=True, download=True)
        mask = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
        oracle.set_mask_threshold_volume(mask, img, (lo, 3071))
        want = oracle.create_surface_piece(None, mask, slice(0, img.shape[0]), (1.0, 1.0, 1.0), 0, 0, True)
        _cmp(got, want)
    vol.close()
