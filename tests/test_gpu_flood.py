"""GPU parity: seeded region growing vs the C oracle (restatement of floodfill.rs), bit-exact."""
import numpy as np
import pytest
from scipy.ndimage import generate_binary_structure

from conftest import synth_volume

pytestmark = pytest.mark.gpu


def test_reference_golden_vectors(ivxlib):
    """tests/test_segmentation_tools.py:17-102 run through the GPU path."""
    from invesalius3_amd import invesalius_rs as floodfill
    image = np.array([[[1, 1, 1, 5, 5], [1, 2, 2, 5, 5], [1, 2, 3, 5, 5], [1, 2, 2, 5, 5], [1, 1, 1, 5, 5]]],
                     dtype=np.int16)
    out_mask = np.zeros((1, 5, 5), dtype=np.uint8)
    floodfill.floodfill_threshold(image, [[2, 2, 0]], 2, 3, 1, generate_binary_structure(3, 1), out_mask)
    expected = np.array([[0, 0, 0, 0, 0], [0, 1, 1, 0, 0], [0, 1, 1, 0, 0], [0, 1, 1, 0, 0], [0, 0, 0, 0, 0]],
                        dtype=np.uint8)
    assert np.array_equal(out_mask[0], expected)
    image = np.array([[[2, 2, 0], [0, 2, 0], [0, 0, 2]]], dtype=np.int16)
    out8 = np.zeros((1, 3, 3), dtype=np.uint8)
    floodfill.floodfill_threshold(image, [[0, 0, 0]], 2, 2, 1, generate_binary_structure(3, 2), out8)
    assert np.array_equal(out8, np.array([[[1, 1, 0], [0, 1, 0], [0, 0, 1]]], dtype=np.uint8))
    out4 = np.zeros((1, 3, 3), dtype=np.uint8)
    floodfill.floodfill_threshold(image, [[0, 0, 0]], 2, 2, 1, generate_binary_structure(3, 1), out4)
    assert np.array_equal(out4, np.array([[[1, 1, 0], [0, 1, 0], [0, 0, 0]]], dtype=np.uint8))


@pytest.mark.parametrize("conn", [1, 2, 3])
@pytest.mark.parametrize("shape", [(24, 40, 70), (17, 33, 129), (40, 48, 64)])
def test_random_volume_matches_oracle(ivxlib, oracle, conn, shape):
    from invesalius3_amd import invesalius_rs as floodfill
    img = synth_volume(shape, seed=41 + conn)
    strct = generate_binary_structure(3, conn)
    z, y, x = np.unravel_index(np.argmax(img), img.shape)
    seeds = [(int(x), int(y), int(z)), (0, 0, 0), (shape[2] - 1, shape[1] - 1, shape[0] - 1)]
    rng = np.random.default_rng(conn)
    out_g = np.where(rng.random(shape) < 0.02, 1, 0).astype(np.uint8)  # pre-filled voxels act as barriers
    out_g[rng.random(shape) < 0.02] = 7
    out_r = out_g.copy()
    t0, t1 = -800.5, 3071.9  # wrapper truncates with int() for integer data
    floodfill.floodfill_threshold(img, seeds, t0, t1, 1, strct, out_g)
    oracle.floodfill_threshold(img, seeds, t0, t1, 1, strct, out_r)
    assert np.array_equal(out_g, out_r)
    assert (out_g == 1).sum() > 1000


def test_noise_maze_many_components(ivxlib, oracle):
    """percolation-like noise: thin, tortuous components crossing many tiles"""
    from invesalius3_amd import invesalius_rs as floodfill
    rng = np.random.default_rng(5)
    for conn, p in ((1, 0.36), (2, 0.16), (3, 0.11)):
        img = (rng.random((40, 70, 150)) < p).astype(np.int16) * 100
        strct = generate_binary_structure(3, conn)
        idx = np.argwhere(img == 100)
        seeds = [tuple(int(v) for v in idx[i][::-1]) for i in rng.integers(0, len(idx), 5)]
        og = np.zeros(img.shape, np.uint8)
        orf = og.copy()
        floodfill.floodfill_threshold(img, seeds, 50, 150, 255, strct, og)
        oracle.floodfill_threshold(img, seeds, 50, 150, 255, strct, orf)
        assert np.array_equal(og, orf)


def test_serpentine_worst_case(ivxlib, oracle):
    """1-voxel-wide corridor snaking through the volume (SURVEY 8d worst case)"""
    from invesalius3_amd import invesalius_rs as floodfill
    dz, dy, dx = 6, 40, 130
    img = np.zeros((dz, dy, dx), np.int16)
    for z in range(0, dz, 2):
        for y in range(0, dy, 2):
            img[z, y, :] = 1
            xs = dx - 1 if (y // 2) % 2 == 0 else 0
            if y + 1 < dy:
                img[z, y + 1, xs] = 1
        if z + 1 < dz:
            last_y = (dy - 1) // 2 * 2
            xs = 0 if (last_y // 2) % 2 == 0 else dx - 1
            img[z + 1, :, :] = 0
            img[z + 1, last_y, xs] = 1
    og = np.zeros(img.shape, np.uint8)
    orf = og.copy()
    s6 = generate_binary_structure(3, 1)
    floodfill.floodfill_threshold(img, [(0, 0, 0)], 1, 1, 1, s6, og)
    oracle.floodfill_threshold(img, [(0, 0, 0)], 1, 1, 1, s6, orf)
    assert np.array_equal(og, orf)
    assert og.sum() > dx * dy // 2


def test_inplace_mask_relabel_on_strided_view(ivxlib, oracle):
    """FloodFillMaskInteractorStyle (styles.py:2517,2534): t in [0,2] -> 254 on mask.matrix[1:,1:,1:];
    RemoveMaskParts (styles.py:2572-2588): t in [253,255] -> 1"""
    from invesalius3_amd import invesalius_rs as floodfill
    img = synth_volume((20, 36, 66), seed=43)
    mg = np.zeros((21, 37, 67), np.uint8)
    mg[1:, 1:, 1:] = np.where(img > -850, 255, 0)
    mr = mg.copy()
    s = generate_binary_structure(3, 1)
    hole = np.argwhere(mg[1:, 1:, 1:] == 0)[0][::-1]
    floodfill.floodfill_threshold_inplace(mg[1:, 1:, 1:], [tuple(int(v) for v in hole)], 0, 2, 254, s)
    oracle.floodfill_threshold_inplace(mr[1:, 1:, 1:], [tuple(int(v) for v in hole)], 0, 2, 254, s)
    assert np.array_equal(mg, mr)
    assert (mg == 254).any()
    part = np.argwhere(mg[1:, 1:, 1:] == 255)[0][::-1]
    s26 = generate_binary_structure(3, 3)
    floodfill.floodfill_threshold_inplace(mg[1:, 1:, 1:], [tuple(int(v) for v in part)], 253, 255, 1, s26)
    oracle.floodfill_threshold_inplace(mr[1:, 1:, 1:], [tuple(int(v) for v in part)], 253, 255, 1, s26)
    assert np.array_equal(mg, mr)
    assert (mg == 1).any()


def test_dtypes_2d_strct_and_errors(ivxlib, oracle):
    from invesalius3_amd import invesalius_rs as floodfill
    rng = np.random.default_rng(8)
    f = rng.normal(0, 1, (9, 20, 70))
    u = (rng.random((9, 20, 70)) * 255).astype(np.uint8)
    s2d = np.zeros((1, 3, 3), np.uint8)
    s2d[0] = [[0, 1, 0], [1, 1, 1], [0, 1, 0]]  # 2-D structuring element (styles.py: 2-D tools)
    for data, t0, t1 in ((f, -0.6, 2.5), (u, 60, 255)):
        idx = np.argwhere((data >= t0) & (data <= t1))[3][::-1]
        seed = [tuple(int(v) for v in idx)]
        for strct in (s2d, generate_binary_structure(3, 3)):
            og = np.zeros(data.shape, np.uint8)
            orf = og.copy()
            floodfill.floodfill_threshold(data, seed, t0, t1, 9, strct, og)
            oracle.floodfill_threshold(data, seed, t0, t1, 9, strct, orf)
            assert np.array_equal(og, orf)
    out = np.zeros((9, 20, 70), np.uint8)
    with pytest.raises(IndexError):
        floodfill.floodfill_threshold(u, [(70, 0, 0)], 0, 255, 1, s2d, out)
    with pytest.raises(TypeError):
        floodfill.floodfill_threshold(u.astype(np.float32), [(0, 0, 0)], 0, 255, 1, s2d, out)
    # uint8 data: thresholds / fill reach the binding's `extract::<u8>()` untouched (invesalius_rs/__init__.py:32-38 only
    # converts for the wider integer dtypes): out of range -> OverflowError, floats -> TypeError
    with pytest.raises(OverflowError):
        floodfill.floodfill_threshold(u, [(0, 0, 0)], 256, 300, 1, s2d, out)
    with pytest.raises(OverflowError):
        floodfill.floodfill_threshold(u, [(0, 0, 0)], 0, 255, 257, s2d, out)
    with pytest.raises(TypeError):
        floodfill.floodfill_threshold(u, [(0, 0, 0)], 0.5, 255, 1, s2d, out)
    # seed value outside the range: nothing happens
    lo = int(u[0, 0, 0]) + 1
    if lo <= 255:
        floodfill.floodfill_threshold(u, [(0, 0, 0)], lo, 255, 1, s2d, out)
    assert out.sum() == 0


def test_full_size_512_properties(ivxlib):
    """512^3: component of the seed equals scipy.ndimage.label's component (26-connectivity); idempotent."""
    from scipy import ndimage
    from invesalius3_amd import invesalius_rs as floodfill
    n = 512
    rng = np.random.default_rng(10)
    small = rng.normal(0, 1, (64, 64, 64))
    img = (ndimage.zoom(small, 8, order=1) * 1000).astype(np.int16)
    strct = generate_binary_structure(3, 3)
    z, y, x = np.unravel_index(np.argmax(img), img.shape)
    out = np.zeros((n, n, n), np.uint8)
    floodfill.floodfill_threshold(img, [(int(x), int(y), int(z))], 200, 32767, 1, strct, out)
    lab, _ = ndimage.label(img >= 200, structure=strct)
    assert np.array_equal(out == 1, lab == lab[z, y, x])
    again = out.copy()
    floodfill.floodfill_threshold(img, [(int(x), int(y), int(z))], 200, 32767, 1, strct, again)
    assert np.array_equal(again, out)


def test_long_serpentine_escapes_to_union_find(ivxlib, oracle):
    """a corridor that needs hundreds of tile hops: the frontier hands over to the union-find path (k_ccl.hip)
    after 48 rounds; the result must still be the oracle's"""
    from invesalius3_amd import invesalius_rs as floodfill
    dz, dy, dx = 4, 400, 200
    img = np.zeros((dz, dy, dx), np.int16)
    for y in range(0, dy, 2):
        img[1, y, :] = 1
        xs = dx - 1 if (y // 2) % 2 == 0 else 0
        if y + 1 < dy:
            img[1, y + 1, xs] = 1
    img[3, ::7, ::5] = 1  # unrelated specks
    for strct in (generate_binary_structure(3, 1), generate_binary_structure(3, 3)):
        og = np.zeros(img.shape, np.uint8)
        orf = og.copy()
        floodfill.floodfill_threshold(img, [(0, 0, 1)], 1, 1, 9, strct, og)
        oracle.floodfill_threshold(img, [(0, 0, 1)], 1, 1, 9, strct, orf)
        assert np.array_equal(og, orf)
        assert (og == 9).sum() >= dx * dy // 2


@pytest.mark.parametrize("mode", ["ccl", "persistent", "rounds", "resident"])
def test_all_flood_engines_agree(ivxlib, oracle, mode):
    """the engines (tile frontier per round, the same rounds in one resident launch, persistent frontier, union-find) in a
    fresh process each"""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import synth_volume\n"
        "from scipy.ndimage import generate_binary_structure\n"
        "from invesalius3_amd import invesalius_rs as ff\n"
        "from oracle import oracle as orc\n"
        "img = synth_volume((40, 72, 136), seed=91)\n"
        "rng = np.random.default_rng(3)\n"
        "for conn in (1, 2, 3):\n"
        "    s = generate_binary_structure(3, conn)\n"
        "    z, y, x = np.unravel_index(np.argmax(img), img.shape)\n"
        "    seeds = [(int(x), int(y), int(z)), (5, 5, 5)]\n"
        "    og = (rng.random(img.shape) < 0.02).astype(np.uint8); orf = og.copy()\n"
        "    ff.floodfill_threshold(img, seeds, -820, 3071, 1, s, og)\n"
        "    orc.floodfill_threshold(img, seeds, -820, 3071, 1, s, orf)\n"
        "    assert np.array_equal(og, orf), conn\n"
        "print('engines-ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                   os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IVX_FLOOD_MODE=mode)
    if mode == "resident":
        env = dict(os.environ, IVX_FLOOD_MODE="rounds", IVX_FLOOD_RESIDENT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "engines-ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("method,use_ww_wl", [("threshold", False), ("dynamic", False), ("dynamic", True), ("confidence", False),
                                               ("confidence", True)])
def test_do_3d_seg_mirror_matches_the_reference_recipe(ivxlib, oracle, method, use_ww_wl):
    """styles.py:3151-3251 end to end: thresholds per method, do_threshold_to_all_slices, flood, mask[out] = fill_value"""
    from scipy.ndimage import generate_binary_structure
    from invesalius3_amd import styles as st
    img = synth_volume((20, 40, 64), seed=91)
    ww, wl = 1500.0, 300.0
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    seed = (int(x), int(y), int(z))
    mask = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    rng = (100, 3071)
    want = mask.copy()
    flood_img = oracle.get_LUT_value_255(img, ww, wl) if use_ww_wl else img
    bstruct = generate_binary_structure(3, 2).astype(np.uint8)
    oracle.do_threshold_to_all_slices(want, img, rng)
    if method == "confidence":
        out = oracle.do_rg_confidence(flood_img, seed, bstruct, 2.5, 3)
    else:
        if method == "threshold":
            t0, t1 = 200, 2500
        else:
            v = int(flood_img[z, y, x])
            t0, t1 = v - 40, v + 60
        out = np.zeros(img.shape, np.uint8)
        oracle.floodfill_threshold(flood_img, [seed], t0, t1, 1, bstruct, out)
    want[1:, 1:, 1:][out.astype(bool)] = 254
    ok = st.do_3d_seg(img, mask, seed, method=method, con_3d=18, fill_value=254, t0=200, t1=2500, dev_min=40, dev_max=60,
                      use_ww_wl=use_ww_wl, ww=ww, wl=wl, threshold_range=rng)
    assert ok and np.array_equal(mask, want)
    assert (mask == 254).sum() == int(out.sum()) > 0
    # a click outside the range is rejected before anything is touched (styles.py:3178)
    if method == "threshold":
        m2 = np.zeros_like(mask)
        assert st.do_3d_seg(img, m2, (0, 0, 0), method="threshold", t0=2000, t1=2500, threshold_range=rng) is False
        assert not m2.any()


@pytest.mark.parametrize("conn", [1, 2, 3])
@pytest.mark.parametrize("shape", [(70, 90, 200), (48, 64, 256), (33, 47, 130)])
def test_solid_bodies_cross_whole_tiles(ivxlib, oracle, conn, shape):
    """Solid bodies made of all-candidate 64x16x16 tiles (the coarse tile-graph pass of k_flood.hip) that touch by a
    face, an edge and a corner only: which of them join depends on the structuring element, tile by tile and voxel by
    voxel alike.  Odd dims leave partial tiles on every high side; pre-filled voxels punch barriers into one body."""
    from invesalius3_amd import invesalius_rs as floodfill
    dz, dy, dx = shape
    img = np.zeros(shape, np.int16)
    img[0:32, 0:32, 0:128] = 500                       # body A: 2x2x2 whole tiles
    img[32:dz, 0:32, 0:128] = 500                      # B: face neighbour of A along z (partial tiles at the far side)
    img[0:16, 32:48, 128:dx] = 500                     # C: touches A by an edge only (y and x both step)
    img[32:48, 32:dy, 128:dx] = 500                    # D: touches A by a corner only
    rng = np.random.default_rng(conn)
    img[(rng.random(shape) < 0.04) & (img == 0)] = 500  # loose voxels around the bodies
    strct = generate_binary_structure(3, conn)
    out_g = np.zeros(shape, np.uint8)
    out_g[8:12, 4:28, 60:70] = 1                       # a pre-filled slab inside A: those tiles are not all-candidate
    out_g[20, :, :] = np.where(rng.random((dy, dx)) < 0.5, 1, 0)
    out_r = out_g.copy()
    seeds = [(100, 20, 28)]
    floodfill.floodfill_threshold(img, seeds, 400, 600, 1, strct, out_g)
    oracle.floodfill_threshold(img, seeds, 400, 600, 1, strct, out_r)
    assert np.array_equal(out_g, out_r)
    assert (out_g == 1).sum() > 32 * 32 * 64
    # a second flood into the same out array from another body: earlier fill acts as barrier, whole tiles now partly blocked
    seeds2 = [(dx - 2, 40, 40)] if dz > 40 else [(dx - 2, 40, 8)]
    floodfill.floodfill_threshold(img, seeds2, 400, 600, 2, strct, out_g)
    oracle.floodfill_threshold(img, seeds2, 400, 600, 2, strct, out_r)
    assert np.array_equal(out_g, out_r)


def test_concurrent_floods_on_separate_streams(ivxlib, oracle):
    """Four resident volumes, each on its own stream and host thread, flood repeatedly at the same time: every stream
    watches its own progress line, and rounds left queued by one flood must not end the next one early."""
    import threading

    from invesalius3_amd.device import DeviceVolume
    strct = generate_binary_structure(3, 3)
    vols, refs, seeds = [], [], []
    for i in range(4):
        img = synth_volume((40 + 8 * i, 48, 128), seed=90 + i)
        z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
        sd = [(int(x), int(y), int(z))]
        ref = np.zeros(img.shape, np.uint8)
        oracle.floodfill_threshold(img, sd, -700, 3071, 1, strct, ref)
        vols.append(DeviceVolume(img))
        refs.append(ref)
        seeds.append(sd)
    errs = []

    def run(i):
        try:
            for rep in range(12):
                vols[i].zero_out_mask()
                vols[i].region_grow(seeds[i], -700, 3071, strct, fill=1, select_value=None)
                if rep % 4 == 3 and not np.array_equal(vols[i].download_out_mask(), refs[i]):
                    errs.append((i, rep))
        except Exception as e:  # pragma: no cover
            errs.append((i, repr(e)))

    th = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    for v in vols:
        v.close()
    assert not errs, errs
    assert all(r.sum() > 1000 for r in refs)


@pytest.mark.parametrize("dtype", [np.int16, np.uint8, np.float64])
def test_floodfill_equal_value_matches_oracle(ivxlib, oracle, dtype):
    """invesalius_rs.floodfill (floodfill.rs:5-49): 6-neighbour component of data == v, seed filled unconditionally."""
    from invesalius3_amd import invesalius_rs as rs
    rng = np.random.default_rng(11)
    shape = (21, 37, 131)
    data = (rng.random(shape) < 0.62).astype(dtype) * (7 if dtype != np.float64 else 7.5)
    v = 7 if dtype != np.float64 else 7.5
    idx = np.argwhere(data == v)
    for n, seed_on_value in enumerate((True, False)):
        z, y, x = idx[rng.integers(len(idx))] if seed_on_value else np.argwhere(data != v)[rng.integers(100)]
        og = (rng.random(shape) < 0.03).astype(np.uint8) * 9  # pre-filled voxels: barriers
        orf = og.copy()
        rs.floodfill(data, int(x), int(y), int(z), v, 9, og)
        oracle.floodfill(data, int(x), int(y), int(z), v, 9, orf)
        assert np.array_equal(og, orf)
        assert og[z, y, x] == 9
        if seed_on_value:
            assert (og == 9).sum() > (orf == 9).sum() * 0 + 1000
    with pytest.raises(IndexError):
        rs.floodfill(data, shape[2], 0, 0, v, 1, np.zeros(shape, np.uint8))
    with pytest.raises(OverflowError):
        rs.floodfill(data, 0, 0, 0, v, 256, np.zeros(shape, np.uint8))
    with pytest.raises(TypeError):
        rs.floodfill(data, 0, 0, 0, v, 1, np.zeros(shape, np.int16))


@pytest.mark.parametrize("p", [0.0, 0.03, 0.2, 1.5, -0.1])
@pytest.mark.parametrize("shape", [(24, 40, 70), (5, 33, 200), (40, 64, 128)])
def test_floodfill_auto_threshold_matches_oracle(ivxlib, oracle, p, shape):
    """invesalius_rs.floodfill_auto_threshold (floodfill_py.rs:12-85): directed reachability, float32 range arithmetic,
    saturating casts; positive, negative and near-saturation values, pre-filled barriers, several seeds (one of them
    on a pre-filled voxel)."""
    from invesalius3_amd import invesalius_rs as rs
    rng = np.random.default_rng(int(abs(p) * 100) + shape[0])
    zz, yy, xx = np.meshgrid(*(np.linspace(-1, 1, s, dtype=np.float32) for s in shape), indexing="ij")
    f = 900.0 * np.exp(-(zz ** 2 + yy ** 2 + xx ** 2) * 2.0) + 120.0 * np.sin(5 * xx) * np.cos(4 * yy) - 60.0
    f += rng.standard_normal(shape).astype(np.float32) * 6.0
    data = f.astype(np.int16)
    data[:, :4, :] = np.where(rng.random((shape[0], 4, shape[2])) < 0.5, 32000, 32767).astype(np.int16)  # saturation band
    data[:, -3:, :] = -30000
    z, y, x = np.unravel_index(int(np.argmax(f)), shape)
    seeds = [(int(x), int(y), int(z)), (0, 0, 0), (shape[2] - 1, shape[1] - 1, shape[0] - 1), (3, 20, 2)]
    og = (rng.random(shape) < 0.02).astype(np.uint8)   # fill = 1: these are barriers
    og[rng.random(shape) < 0.02] = 5                    # other values are not
    og[2, 20, 3] = 1                                     # a seed on a pre-filled voxel is still expanded
    orf = og.copy()
    rs.floodfill_auto_threshold(data, seeds, p, 1, og)
    oracle.floodfill_auto_threshold(data, seeds, p, 1, orf)
    assert np.array_equal(og, orf)
    if p == 0.2:
        assert (og == 1).sum() > 0.02 * og.size + 1000


def test_floodfill_auto_threshold_on_strided_views_and_errors(ivxlib, oracle):
    from invesalius3_amd import invesalius_rs as rs
    rng = np.random.default_rng(2)
    big = (rng.integers(97, 106, (30, 41, 90))).astype(np.int16)  # steps of up to 8 against a range of about +-4
    data = big[1:, 1:, 1:]
    out_big = np.zeros((30, 41, 90), np.uint8)
    og, orf = out_big[1:, 1:, 1:], np.zeros(data.shape, np.uint8)
    rs.floodfill_auto_threshold(data, [[5, 5, 5]], 0.04, 200, og)
    oracle.floodfill_auto_threshold(np.ascontiguousarray(data), [(5, 5, 5)], 0.04, 200, orf)
    assert np.array_equal(og, orf) and out_big[0].sum() == 0 and (og == 200).sum() > 100
    with pytest.raises(TypeError):
        rs.floodfill_auto_threshold(data.astype(np.uint8), [(0, 0, 0)], 0.1, 1, og)
    with pytest.raises(IndexError):
        rs.floodfill_auto_threshold(data, [(0, 0, 29)], 0.1, 1, og)


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("shape", [(32, 32, 32), (20, 37, 70), (1, 40, 65), (64, 64, 64)])
def test_jump_flooding_matches_oracle(ivxlib, oracle, shape, normalize):
    """invesalius_rs.jump_flooding (floodfill.rs:298-507): owners exactly, float32 distances bit for bit (sqrt and the
    normalising division are correctly rounded on both sides); sites outside the volume, two sites on one voxel,
    pre-set owners / distances in the input arrays."""
    from invesalius3_amd import invesalius_rs as rs
    rng = np.random.default_rng(shape[1] + int(normalize))
    n = 23
    sites = np.stack([rng.integers(0, shape[0], n), rng.integers(0, shape[1], n), rng.integers(0, shape[2], n)], 1).astype(np.int32)
    sites[3] = (-1, 5, 5)                 # skipped when seeding
    sites[4] = (shape[0], 0, 0)           # out of bounds: skipped
    sites[7] = sites[6]                   # two sites on one voxel: the later one owns it
    dg = np.full(shape, -1.0, np.float32)
    og = np.zeros(shape, np.int32)
    og[0, 3, 4], dg[0, 3, 4] = 2, 0.25    # an owner handed in by the caller, with a distance that is not the true one
    og[0, 9, 9] = n + 5                   # refers to no site: never adopted by anybody
    dr, orf = dg.copy(), og.copy()
    rs.jump_flooding(dg, og, sites, normalize)
    oracle.jump_flooding(dr, orf, sites, normalize)
    assert np.array_equal(og, orf)
    assert np.array_equal(dg.view(np.uint32), dr.view(np.uint32))
    assert (og > 0).mean() > 0.99
    if normalize:
        assert float(dg[og > 0].max()) <= 1.0


def test_jump_flooding_views_errors_and_empty(ivxlib, oracle):
    from invesalius3_amd import invesalius_rs as rs
    big_d = np.full((12, 20, 40), -1.0, np.float32)
    big_o = np.zeros((12, 20, 40), np.int32)
    d, o = big_d[1:, 2:, 3:], big_o[1:, 2:, 3:]
    sites = np.array([[2, 3, 4], [8, 10, 30]], np.int32)
    rs.jump_flooding(d, o, sites, False)
    dr, orf = np.full(d.shape, -1.0, np.float32), np.zeros(d.shape, np.int32)
    oracle.jump_flooding(dr, orf, sites, False)
    assert np.array_equal(o, orf) and np.array_equal(d, dr) and big_o[0].sum() == 0 and (big_d[:, :2] == -1).all()
    rs.jump_flooding(d, o, np.zeros((0, 3), np.int32), True)   # no sites: nothing happens (floodfill.rs:310-312)
    assert np.array_equal(o, orf)
    with pytest.raises(TypeError):
        rs.jump_flooding(d.astype(np.float64), o, sites, False)
    with pytest.raises(TypeError):
        rs.jump_flooding(d, o, sites.astype(np.int64), False)


def test_jump_flooding_many_sites_normalised(ivxlib, oracle):
    """more sites than the workgroup-level accumulators hold (1024): the global-atomic path of the normalisation"""
    from invesalius3_amd import invesalius_rs as rs
    shape = (24, 48, 96)
    rng = np.random.default_rng(9)
    n = 1500
    sites = np.stack([rng.integers(0, shape[0], n), rng.integers(0, shape[1], n), rng.integers(0, shape[2], n)], 1).astype(np.int32)
    dg, og = np.full(shape, -1.0, np.float32), np.zeros(shape, np.int32)
    dr, orf = dg.copy(), og.copy()
    rs.jump_flooding(dg, og, sites, True)
    oracle.jump_flooding(dr, orf, sites, True)
    assert np.array_equal(og, orf)
    assert np.array_equal(dg.view(np.uint32), dr.view(np.uint32))


def test_more_than_64_resident_volumes_can_flood(ivxlib, oracle):
    """every resident volume owns a stream and every stream a progress line: the pool of lines grows on demand"""
    from invesalius3_amd.device import DeviceVolume
    strct = generate_binary_structure(3, 1)
    img = synth_volume((8, 16, 64), seed=5)
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    ref = np.zeros(img.shape, np.uint8)
    oracle.floodfill_threshold(img, [(int(x), int(y), int(z))], -900, 3071, 1, strct, ref)
    vols = [DeviceVolume(img) for _ in range(70)]
    try:
        for v in vols:
            v.region_grow([(int(x), int(y), int(z))], -900, 3071, strct, fill=1, select_value=None)
        assert all(np.array_equal(v.download_out_mask(), ref) for v in vols[::7])
    finally:
        for v in vols:
            v.close()


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_seed_on_a_tile_face_reaches_the_next_tile(ivxlib, oracle, axis):
    """A seed whose only in-range neighbours lie in the ADJACENT flood tile (tiles are 64 x 16 x 16 voxels): the seed's
    own tile visit gains nothing, so the neighbours have to be woken by the seeding itself (regression: found by the
    8-slab configs[3] test, where a plane OR-ed into a halo slice hit the same blind spot)."""
    from invesalius3_amd import invesalius_rs as rs
    shape = (40, 40, 160)
    img = np.zeros(shape, np.int16)
    lo = [5, 5, 5]
    lo[axis] = {0: 15, 1: 15, 2: 63}[axis]          # last voxel of the first tile along `axis`
    hi = list(lo)
    hi[axis] += 20
    sl = tuple(slice(a, b + 1) for a, b in zip(lo, hi))
    img[sl] = 1000                                   # a 1-voxel-wide line leaving the seed's tile
    seed = (lo[2], lo[1], lo[0])
    for conn in (1, 3):
        s = generate_binary_structure(3, conn)
        g, r = np.zeros(shape, np.uint8), np.zeros(shape, np.uint8)
        rs.floodfill_threshold(img, [seed], 500, 2000, 1, s, g)
        oracle.floodfill_threshold(img, [seed], 500, 2000, 1, s, r)
        assert r.sum() == 21 and np.array_equal(g, r)


def test_resident_launch_behind_the_async_grow_of_the_resident_pipeline(ivxlib, oracle):
    """IVX_FLOOD_RESIDENT=1 in a fresh process: DeviceVolume.region_grow returns right behind the resident launch
    (ivx_dev_flood_grow_async), queues `mask[reached] = 254` at once and fetches the round count afterwards
    (ivx_dev_flood_wait): same mask, same out_mask, a positive round count"""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import synth_volume\n"
        "from scipy.ndimage import generate_binary_structure\n"
        "from invesalius3_amd.device import DeviceVolume\n"
        "from oracle import oracle as orc\n"
        "img = synth_volume((40, 72, 136), seed=93)\n"
        "s26 = generate_binary_structure(3, 3).astype(np.uint8)\n"
        "lo, hi = 150, 3071\n"
        "z, y, x = np.unravel_index(np.argmax(np.where(img <= hi, img, -4000)), img.shape)\n"
        "vol = DeviceVolume(img, spacing=(1.0, 1.0, 1.0))\n"
        "for _ in range(2):\n"
        "    vol.zero_out_mask(); vol.threshold(lo, hi)\n"
        "    rounds = vol.region_grow([(int(x), int(y), int(z))], lo, hi, s26, fill=1, select_value=254)\n"
        "    assert rounds >= 1, rounds\n"
        "mask = np.where((img >= lo) & (img <= hi), 255, 0).astype(np.uint8)\n"
        "out = np.zeros(img.shape, np.uint8)\n"
        "orc.floodfill_threshold(img, [(int(x), int(y), int(z))], lo, hi, 1, s26, out)\n"
        "mask[out.astype(bool)] = 254\n"
        "assert np.array_equal(vol.download_mask(), mask) and np.array_equal(vol.download_out_mask(), out)\n"
        "assert (mask == 254).sum() > 1000\n"
        "print('resident-ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IVX_FLOOD_RESIDENT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "resident-ok" in r.stdout, r.stdout + r.stderr
