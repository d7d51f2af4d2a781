"""GPU parity: fill holes, window/level LUT, min-shift, morphological gradient, watershed merge, confidence region
growing and the resident DeviceVolume pipeline -- against numpy / scipy / the C oracle, bit-exact."""
import numpy as np
import pytest
from scipy import ndimage
from scipy.ndimage import generate_binary_structure

from conftest import synth_volume

pytestmark = pytest.mark.gpu


def test_fill_holes_reference_fixture(ivxlib):
    """tests/test_segmentation_tools.py:105-134 through the GPU path."""
    from invesalius3_amd import invesalius_rs as floodfill
    mask_2d = np.ones((7, 7), dtype=np.uint8)
    mask_2d[3, 3] = 0
    mask = mask_2d[np.newaxis, ...].copy()
    labels_2d, _ = ndimage.label(mask_2d == 0, structure=np.ones((3, 3), dtype=np.uint8), output=np.uint32)
    labels = labels_2d[np.newaxis, ...]
    ret = floodfill.fill_holes_automatically(mask, labels, int(labels.max()), 1)
    expected = np.ones((1, 7, 7), dtype=np.uint8)
    expected[0, 3, 3] = 254
    assert ret and np.array_equal(mask, expected)


@pytest.mark.parametrize("max_size", [0, 3, 50, 10 ** 9])
def test_fill_holes_matches_oracle(ivxlib, oracle, max_size):
    """Mask.fill_holes_auto (mask.py:519-562): label the background, fill the small components."""
    from invesalius3_amd import invesalius_rs as floodfill
    img = synth_volume((30, 50, 70), seed=61)
    big = np.zeros((31, 51, 71), np.uint8)
    big[1:, 1:, 1:] = np.where(img > -820, 255, 0)
    mg = big.copy()
    mr = big.copy()
    labels, nlabels = ndimage.label(big[1:, 1:, 1:] < 127, structure=generate_binary_structure(3, 1), output=np.uint32)
    rg = floodfill.fill_holes_automatically(mg[1:, 1:, 1:], labels, nlabels, max_size)
    rr = oracle.fill_holes_automatically(mr[1:, 1:, 1:], labels, nlabels, max_size)
    assert rg == rr
    assert np.array_equal(mg, mr)
    with pytest.raises(IndexError):
        floodfill.fill_holes_automatically(mg[1:, 1:, 1:], labels, max(int(nlabels) - 1, 0), 5)


@pytest.mark.parametrize("ww,wl", [(400, 300), (2000, 500), (255, 127), (80.5, 40.25)])
def test_lut_and_minshift_cost_images(ivxlib, oracle, ww, wl):
    from invesalius3_amd import watershed_process as wp
    img = synth_volume((12, 33, 47), seed=62)
    exp = oracle.get_LUT_value(img, ww, wl).astype("uint16")  # watershed_process.py:34
    assert np.array_equal(wp.cost_image(img, True, wl, ww), exp)
    exp2 = (img - img.min()).astype("uint16")  # watershed_process.py:47
    assert np.array_equal(wp.cost_image(img, False, wl, ww), exp2)
    sl = img[5]  # 2-D variant (styles.py:1926-2000)
    assert np.array_equal(wp.cost_image(sl, True, wl, ww), oracle.get_LUT_value(sl, ww, wl).astype("uint16"))


@pytest.mark.parametrize("size", [1, 3, 5])
def test_morphological_gradient_matches_scipy(ivxlib, oracle, size):
    from invesalius3_amd import watershed_process as wp
    img = synth_volume((9, 21, 40), seed=63)
    lut = oracle.get_LUT_value(img, 600, 200).astype("uint16")
    exp = ndimage.morphological_gradient(lut, size)  # watershed_process.py:36-38
    assert exp.dtype == np.uint16
    assert np.array_equal(wp.cost_image(img, True, 200, 600, size), exp)
    exp2 = ndimage.morphological_gradient((img - img.min()).astype("uint16"), size)
    assert np.array_equal(wp.cost_image(img, False, 0, 0, size), exp2)
    aniso = (1, 3, 5)
    assert np.array_equal(wp.cost_image(img, True, 200, 600, aniso), ndimage.morphological_gradient(lut, aniso))
    if size == 3:
        sl = img[4]
        assert np.array_equal(wp.cost_image(sl, False, 0, 0, 3), ndimage.morphological_gradient((sl - sl.min()).astype("uint16"), 3))


@pytest.mark.parametrize("shape", [(70, 21, 40), (33, 8, 8), (4, 5, 16), (3, 9, 24), (1, 7, 8)])
def test_morphological_gradient_3_over_slice_segments(ivxlib, oracle, shape, monkeypatch):
    """The 3x3x3 gradient walks along z in segments of 32 slices with three slices' in-slice extrema in registers (k_misc.hip):
    segment seams, a ragged last segment, volumes thinner than the window, both kernels (IVX_MG_WALK is read once per process:
    the other one is reached through thin volumes) == scipy's morphological_gradient bit for bit."""
    from invesalius3_amd import watershed_process as wp
    img = synth_volume(shape, seed=64)
    img[::7, ::3, ::5] = 3000
    img[3::11, 1::4, 2::9] = -1000
    base = (img - img.min()).astype("uint16")
    assert np.array_equal(wp.cost_image(img, False, 0, 0, 3), ndimage.morphological_gradient(base, 3))
    lut = oracle.get_LUT_value(img, 600, 200).astype("uint16")
    assert np.array_equal(wp.cost_image(img, True, 200, 600, 3), ndimage.morphological_gradient(lut, 3))


@pytest.mark.parametrize("overwrite", [False, True])
def test_watershed_merge_rule(ivxlib, oracle, overwrite):
    from invesalius3_amd import watershed_process as wp
    rng = np.random.default_rng(64)
    big = rng.choice(np.array([0, 1, 2, 253, 254, 255, 7], np.uint8), (11, 21, 31))
    tmp = rng.integers(0, 3, (10, 20, 30)).astype(np.uint8)
    mg, mr = big.copy(), big.copy()
    wp.merge(mg[1:, 1:, 1:], tmp, overwrite)
    oracle.watershed_merge(mr[1:, 1:, 1:], tmp, overwrite)
    assert np.array_equal(mg, mr)


def test_do_watershed_ift_pipeline_matches_reference_calls(ivxlib, tmp_path):
    """tests/test_segmentation_tools.py:170-213 analogue for the IFT branch: same memmap + queue protocol, cost image
    and marker flood on the GPU, labels equal the reference pipeline's (live scipy), with and without ww/wl."""
    import queue
    from invesalius3_amd import watershed_process as wp
    image = np.zeros((5, 5, 5), dtype=np.int16)
    image[1:4, 1:4, 1:4] = 100
    markers = np.zeros((5, 5, 5), dtype=np.int16)
    markers[2, 2, 2] = 1
    markers[0, 0, 0] = 2
    tfile = str(tmp_path / "ws.dat")
    np.memmap(tfile, shape=image.shape, dtype="uint8", mode="w+").flush()
    q = queue.Queue()
    bstruct = generate_binary_structure(3, 1)
    wp.do_watershed(image, markers, tfile, image.shape, bstruct, "Watershed IFT", (3, 3, 3), False, 0, 0, q)
    assert q.get() == 1
    got = np.array(np.memmap(tfile, shape=image.shape, dtype="uint8", mode="r"))
    exp = ndimage.watershed_ift((image - image.min()).astype("uint16"), markers.astype("int8"), bstruct)
    assert np.array_equal(got, exp.astype(np.uint8))
    assert (got == 1).sum() == 27 and (got == 2).sum() == 98  # SURVEY 8c golden counts
    wp.do_watershed(image, markers, tfile, image.shape, bstruct, "Watershed IFT", (3, 3, 3), True, 50, 120, q)
    assert q.get() == 1
    got = np.array(np.memmap(tfile, shape=image.shape, dtype="uint8", mode="r"))
    lut = np.piecewise(image, [image <= (50 - 0.5 - (120 - 1) / 2.0), image > (50 - 0.5 + (120 - 1) / 2.0)],
                       [0, 120, lambda v: ((v - (50 - 0.5)) / (120 - 1) + 0.5) * 120])
    exp = ndimage.watershed_ift(lut.astype("uint16"), markers.astype("int16"), bstruct)
    assert np.array_equal(got, exp.astype(np.uint8))
    # the scikit-image branch (the GUI's default): gradient image + heap flood; counts from live scikit-image (tests/golden)
    wp.do_watershed(image, markers, tfile, image.shape, bstruct, "Watershed", (3, 3, 3), False, 0, 0, q)
    assert q.get() == 1
    got = np.array(np.memmap(tfile, shape=image.shape, dtype="uint8", mode="r"))
    assert (got == 1).sum() == 109 and (got == 2).sum() == 16


def test_device_volume_pipeline_matches_oracle(ivxlib, oracle):
    """resident pipeline == host pipeline == oracle: threshold -> 26-conn region grow -> select -> MC"""
    from invesalius3_amd.device import DeviceVolume
    img = synth_volume((40, 64, 96), seed=65)
    strct = generate_binary_structure(3, 3)
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    seed = (int(x), int(y), int(z))
    vol = DeviceVolume(img, spacing=(0.5, 0.5, 2.0))
    vol.threshold(-700, 3071)
    vol.region_grow([seed], -700, 3071, strct, fill=1, select_value=254)
    tri = vol.marching_cubes(from_binary=True, download=True)
    m_ref = np.zeros((41, 65, 97), np.uint8)
    oracle.set_mask_threshold_volume(m_ref, img, (-700, 3071))
    out_ref = np.zeros(img.shape, np.uint8)
    oracle.floodfill_threshold(img, [seed], -700, 3071, 1, strct, out_ref)
    m_ref[1:, 1:, 1:][out_ref.astype(bool)] = 254
    assert np.array_equal(vol.download_out_mask(), out_ref)
    assert np.array_equal(vol.download_mask(), m_ref[1:, 1:, 1:])
    assert vol.reached_count() == int(out_ref.sum())
    tri_ref = oracle.create_surface_piece(None, m_ref, slice(0, 40), (0.5, 0.5, 2.0), 0, 0, True)
    assert tri.shape == tri_ref.shape and np.array_equal(tri, tri_ref)
    # two-iso "Default" mode on the resident image
    tri2 = vol.marching_cubes(from_binary=False, min_value=-700, max_value=3071, download=True)
    tri2_ref = oracle.create_surface_piece(img, m_ref, slice(0, 40), (0.5, 0.5, 2.0), -700, 3071, False)
    assert np.array_equal(tri2, tri2_ref)
    vol.close()


def test_confidence_region_growing_matches_oracle(ivxlib, oracle):
    """do_rg_confidence (styles.py:3220-3251) incl. quirk Q4 (out_mask never cleared between iterations)"""
    from invesalius3_amd.device import DeviceVolume
    img = synth_volume((32, 48, 80), seed=66)
    strct = generate_binary_structure(3, 2)
    for pick in (np.argmax(img), img.size // 2 + 1234):
        z, y, x = np.unravel_index(int(pick), img.shape)
        vol = DeviceVolume(img)
        vol.region_grow_confidence((int(x), int(y), int(z)), strct, 2.5, 3, select_value=254)
        ref = oracle.do_rg_confidence(img, (int(x), int(y), int(z)), strct, 2.5, 3)
        assert np.array_equal(vol.download_out_mask(), ref)
        exp_mask = np.zeros(img.shape, np.uint8)
        exp_mask[ref.astype(bool)] = 254
        assert np.array_equal(vol.download_mask(), exp_mask)
        vol.close()


def test_dynamic_and_confidence_region_growing_on_lut_image(ivxlib, oracle):
    """styles.py:3166-3178 ("dynamic": v +- dev on get_LUT_value_255(image)) and 3222-3225 (confidence + use_ww_wl)"""
    from invesalius3_amd.device import DeviceVolume
    img = synth_volume((28, 44, 72), seed=67)
    ww, wl = 1200, 100
    lut = oracle.get_LUT_value_255(img, ww, wl)
    assert lut.dtype == np.int16
    strct = generate_binary_structure(3, 3)
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    seed = (int(x), int(y), int(z))
    vol = DeviceVolume(img)
    d_lut = vol.lut_image_255(ww, wl)
    vol.sync()
    assert np.array_equal(d_lut.download(img.shape, np.int16), lut)
    v = int(lut[z, y, x])
    vol.region_grow([seed], v - 60, v + 10, strct, fill=1, select_value=254, image=d_lut)
    ref = np.zeros(img.shape, np.uint8)
    oracle.floodfill_threshold(lut, [seed], v - 60, v + 10, 1, strct, ref)
    assert np.array_equal(vol.download_out_mask(), ref) and ref.sum() > 100
    vol.out_mask.zero(vol.stream)
    vol.region_grow_confidence(seed, strct, 2.5, 3, select_value=None, image=d_lut)
    assert np.array_equal(vol.download_out_mask(), oracle.do_rg_confidence(lut, seed, strct, 2.5, 3))
    d_lut.close()
    vol.close()


@pytest.mark.parametrize("op", [1, 2, 3, 4])
def test_mask_boolean_is_the_references_numpy_expression(ivxlib, op):
    """Slice.do_boolean_op, slice_.py:1906-1916"""
    from invesalius3_amd import slice_ as sl
    rng = np.random.default_rng(op)
    vals = np.array([0, 1, 2, 3, 127, 253, 254, 255], np.uint8)
    m1 = rng.choice(vals, size=(12, 21, 70))
    m2 = rng.choice(vals, size=(12, 21, 70))
    a, b = m1[1:, 1:, 1:], m2[1:, 1:, 1:]
    want = {1: ((a > 2) + (b > 2)) * 255, 2: ((a > 2) ^ ((a > 2) & (b > 2))) * 255, 3: ((a > 2) & (b > 2)) * 255,
            4: np.logical_xor((a > 2), (b > 2)) * 255}[op].astype(np.uint8)
    got = sl.do_boolean_op(op, m1, m2)
    assert got.shape == m1.shape and np.array_equal(got[1:, 1:, 1:], want)
    assert (got[0] == 1).all() and (got[:, 0] == 1).all() and (got[:, :, 0] == 1).all()  # future_mask.matrix[:] = 1
    with pytest.raises(ValueError):
        sl.do_boolean_op(9, m1, m2)


def test_calc_image_density_matches_numpy(ivxlib):
    """Slice.calc_image_density, slice_.py:2284-2297"""
    from invesalius3_amd import slice_ as sl
    img = synth_volume((20, 33, 65), seed=71)
    rng = np.random.default_rng(2)
    m = np.zeros((21, 34, 66), np.uint8)
    m[1:, 1:, 1:] = rng.choice(np.array([0, 127, 128, 254, 255], np.uint8), size=img.shape)
    vals = img[m[1:, 1:, 1:] > 127]
    lo, hi, mean, std = sl.calc_image_density(img, m)
    assert (lo, hi) == (int(vals.min()), int(vals.max()))
    assert mean == pytest.approx(float(vals.mean()), rel=1e-13) and std == pytest.approx(float(vals.std()), rel=1e-10)
    assert sl.calc_image_density(img, np.zeros_like(m)) == (0, 0, 0, 0)


@pytest.mark.parametrize("use_ww_wl,overwrite", [(False, True), (True, False)])
def test_device_volume_watershed_matches_reference_recipe(ivxlib, oracle, use_ww_wl, overwrite):
    """resident watershed == cost image of watershed_process.py:41-57 -> flood (defect-free oracle; == live scipy when its
    unlink defect stays harmless) -> merge rule of styles.py:2147-2152"""
    from invesalius3_amd.device import DeviceVolume
    img = synth_volume((24, 48, 64), seed=66)
    mk = np.zeros(img.shape, np.int16 if use_ww_wl else np.int8)
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    mk[z, y, x] = 1
    mk[0, 0, 0] = mk[-1, -1, -1] = 2
    s = generate_binary_structure(3, 1)
    vol = DeviceVolume(img)
    vol.threshold(226, 3071)
    before = vol.download_mask()
    vol.watershed(mk, s, use_ww_wl=use_ww_wl, wl=300, ww=400, overwrite=overwrite)
    if use_ww_wl:
        cost = np.piecewise(img, [img <= (300 - 0.5 - (400 - 1) / 2.0), img > (300 - 0.5 + (400 - 1) / 2.0)],
                            [0, 400, lambda v: ((v - (300 - 0.5)) / (400 - 1) + 0.5) * 400]).astype(np.uint16)
    else:
        cost = (img - img.min()).astype(np.uint16)
    lab = oracle.watershed_ift_clean(cost, mk, s).astype(np.uint8)
    want = before.copy()
    oracle.watershed_merge(want, lab, overwrite)
    assert np.array_equal(vol.download_mask(), want) and (want == 253).any()
    assert overwrite or (want == 2).any()  # overwrite keeps the object only (styles.py:2147-2149)
    vol.close()


def test_device_volume_watershed_default_algorithm(ivxlib, oracle):
    """resident watershed, algorithm "Watershed" (the GUI's default): min-shift -> 3x3x3 gradient -> scikit-image's flood
    (serial heap flood of oracle/) -> merge rule"""
    from scipy import ndimage

    from invesalius3_amd.device import DeviceVolume
    img = synth_volume((24, 48, 64), seed=67)
    mk = np.zeros(img.shape, np.int16)
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    mk[max(z - 1, 0):z + 2, y - 2:y + 3, x - 2:x + 3] = 1
    mk[:2, :4, :4] = 2
    mk[-2:, -4:, -4:] = 2
    s = generate_binary_structure(3, 1)
    vol = DeviceVolume(img)
    vol.threshold(226, 3071)
    before = vol.download_mask()
    stats = vol.watershed(mk, s, overwrite=False, algorithm="Watershed", mg_size=(3, 3, 3))
    grad = ndimage.morphological_gradient((img - img.min()).astype("uint16"), (3, 3, 3))  # watershed_process.py:49-51
    lab = oracle.watershed_sk(grad, mk, s, 1)
    assert np.array_equal(lab, oracle.watershed_sk(grad, mk, s, 0))  # (heap-ordered marker ties give the same here)
    want = before.copy()
    oracle.watershed_merge(want, lab.astype(np.uint8), False)
    got = vol.download_mask()
    # (the object's voxels hold 255 from the threshold here, which the merge rule leaves alone: styles.py:2150-2152)
    assert np.array_equal(got, want) and (want == 2).any() and (got != before).any()
    assert stats["markers"] == int((mk != 0).sum())
    vol.close()


def test_pageable_copies_through_the_lane_buffers_round_trip(ivxlib, tmp_path):
    """ivx_memcpy_h2d / _d2h with pageable host memory (numpy arrays, np.memmap: what the reference hands over) go through
    page-locked lane buffers on several threads and streams (csrc/ivx_runtime.hip staged_copy): every byte arrives, for sizes
    around the chunk and lane boundaries, unaligned host pointers, a read-only memmap source and a fresh destination; a
    page-locked array (ivx_host_alloc) takes the direct path and must agree."""
    from invesalius3_amd import _lib as L
    from invesalius3_amd.device import DeviceBuffer
    rng = np.random.default_rng(5)
    chunk = 4 << 20
    for n in (chunk - 1, chunk, chunk + 3, 2 * chunk + 1, 7 * chunk + 12345, 13 * chunk, 37 * chunk + 5):
        src = rng.integers(0, 256, n + 7, dtype=np.uint8)[3:3 + n]  # (an unaligned view)
        d = DeviceBuffer(n)
        d.upload(src)
        back = np.empty(n + 5, np.uint8)[5:]
        assert np.array_equal(d.download((n,), np.uint8, out=np.ascontiguousarray(back)), src), n
        pin = L.pinned_empty((n,), np.uint8)
        d.download((n,), np.uint8, out=pin)
        assert np.array_equal(pin, src), n
        pin[:] = pin[::-1].copy()
        d.upload(pin)
        assert np.array_equal(d.download((n,), np.uint8), src[::-1]), n
        d.close()
    n = 9 * chunk + 77
    m = np.memmap(str(tmp_path / "v.dat"), dtype=np.uint8, mode="w+", shape=(n,))
    m[:] = rng.integers(0, 256, n, dtype=np.uint8)
    m.flush()
    ro = np.memmap(str(tmp_path / "v.dat"), dtype=np.uint8, mode="r", shape=(n,))
    d = DeviceBuffer(n)
    d.upload(ro)
    out = np.memmap(str(tmp_path / "o.dat"), dtype=np.uint8, mode="w+", shape=(n,))
    d.download((n,), np.uint8, out=out)
    assert np.array_equal(out, ro)
    d.close()
