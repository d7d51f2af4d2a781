"""Full-size parity gates on the BASELINE configs, through the paths bench.py times (VERDICT r1 "next round" item 2):

* V512 of bench.py through the resident DeviceVolume step -- fused, IVX_NO_FUSE=1 and with the marching-cubes prefetch --
  mask, out_mask and the float32 triangle soup equal the CPU oracle bit for bit (512^3, ~6.3 M triangles);
* MIDA / LMIP / contour MIP at 512^3 exact against the C oracle (restatement of mips.rs; unpinned upstream: no test there);
* configs[3] geometry: 8 loop-back ranks x 32 slices (2 tiles deep) x 2048^2: sharded threshold + region growing +
  marching cubes, the DEVICE stitch of the ranks' indexed pieces == its numpy restatement == the single-volume indexed mesh.
"""
import os
import threading
import zlib

import numpy as np
import pytest
from scipy.ndimage import generate_binary_structure

pytestmark = pytest.mark.gpu
S26 = generate_binary_structure(3, 3)
BONE = (226, 3071)


def _full_fixture(n, img):
    """tests/golden/ws{n}_full.npz when it was made on exactly this synthetic volume (bench.full_volume_fixture)"""
    from bench import full_volume_fixture
    return full_volume_fixture(n, img)


@pytest.fixture(scope="module")
def v512():
    from bench import synth_v512
    img = synth_v512((512, 512, 512))
    z, y, x = np.unravel_index(int(np.argmax(img)), img.shape)
    return img, (int(x), int(y), int(z))


@pytest.fixture(scope="module")
def v512_oracle(v512, oracle):
    """threshold -> 26-neighbour flood from the brightest voxel -> mask[reached] = 254 -> soup over the reference's pieces"""
    from concurrent.futures import ThreadPoolExecutor
    img, seed = v512
    mask = np.zeros(tuple(s + 1 for s in img.shape), np.uint8)
    oracle.set_mask_threshold_volume(mask, img, BONE)
    out = np.zeros(img.shape, np.uint8)
    oracle.floodfill_threshold(img, [seed], BONE[0], BONE[1], 1, S26, out)
    mask[1:, 1:, 1:][out.astype(bool)] = 254
    rois = [slice(i * 20, i * 20 + 21) for i in range(26)]
    with ThreadPoolExecutor(8) as pool:
        parts = list(pool.map(lambda r: oracle.create_surface_piece(None, mask, r, (1.0, 1.0, 1.0), 0, 0, True), rois))
    return np.ascontiguousarray(mask[1:, 1:, 1:]), out, np.concatenate(parts)


@pytest.mark.parametrize("mode", ["fused", "nofuse", "prefetch"])
def test_v512_bench_step_equals_oracle_bit_for_bit(ivxlib, v512, v512_oracle, monkeypatch, mode):
    from invesalius3_amd.device import DeviceVolume
    img, seed = v512
    mask0, out0, soup0 = v512_oracle
    if mode == "nofuse":
        monkeypatch.setenv("IVX_NO_FUSE", "1")
    vol = DeviceVolume(img)
    for _ in range(2):  # bench.py's step, twice: the second one starts from the first one's state
        vol.zero_out_mask()
        vol.threshold(BONE[0], BONE[1], preserve=False)
        if mode == "prefetch":
            vol.surface_prefetch(from_binary=True)
        vol.region_grow([seed], BONE[0], BONE[1], S26, fill=1, select_value=254)
        ntri = vol.marching_cubes(from_binary=True)
    assert ntri == len(soup0) == 6323604 and vol.reached_count() == int(out0.sum()) == 19797285
    soup = vol.marching_cubes(from_binary=True, download=True)
    assert soup.shape == soup0.shape and np.array_equal(soup.view(np.uint32), soup0.view(np.uint32)), "triangle soup"
    assert np.array_equal(vol.download_mask(), mask0), "mask"
    assert np.array_equal(vol.download_out_mask(), out0), "out_mask"
    vol.close()


def test_v512_strong_scaling_split_eight_slabs_of_64_slices_equals_the_single_volume(ivxlib, v512, v512_oracle):
    """`bench.py --gpus 8 --scaling strong`: configs[1]'s ONE 512^3 volume as 8 Z-slabs of 64 slices (4 flood-tile layers each), one
    seed that lies in one slab only; 8 loop-back ranks on this GPU through the real sharded HIP path (image halo, plane exchange
    + vote per round, per-rank marching-cubes piece).  Concatenated masks, out_masks and the triangle soup == the oracle's
    whole-volume bits -- the same arrays the default line's parity gate checks."""
    from _ptr_comm import LoopbackWorld
    from invesalius3_amd.parallel import SlabVolume

    img, seed = v512
    mask0, out0, soup0 = v512_oracle
    world, nz = 8, 64
    lw = LoopbackWorld(world)
    res, errs = {}, []

    def run(rank):
        try:
            vol = SlabVolume(img[rank * nz:(rank + 1) * nz], rank, world, comm=lw.comm(rank))
            for _ in range(2):  # bench.py's step, twice
                vol.zero_out_mask()
                vol.threshold(BONE[0], BONE[1], preserve=False)
                vol.region_grow([seed], BONE[0], BONE[1], S26, fill=1, select_value=254)
                ntri = vol.marching_cubes(from_binary=True)
            lay = vol.lay
            res[rank] = dict(ntri=ntri, count=vol.reached_count(), tris=vol.marching_cubes(from_binary=True, download=True),
                             mask=vol.download_mask()[lay.first_interior:lay.last_interior + 1],
                             out=vol.download_out_mask()[lay.first_interior:lay.last_interior + 1])
            vol.close()
        except Exception:  # pragma: no cover
            import traceback
            errs.append((rank, traceback.format_exc()))
            lw.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=900) for t in th]
    assert not errs, errs
    assert sum(res[r]["count"] for r in range(world)) == int(out0.sum()) == 19797285
    assert np.array_equal(np.concatenate([res[r]["mask"] for r in range(world)]), mask0), "mask"
    assert np.array_equal(np.concatenate([res[r]["out"] for r in range(world)]), out0), "out_mask"
    cat = np.concatenate([res[r]["tris"] for r in range(world)])
    assert sum(res[r]["ntri"] for r in range(world)) == len(cat) == len(soup0) == 6323604
    # (as a multiset: the ranks' pieces are cut at other slices than the oracle's 20-slice pieces, so the order differs)
    assert np.array_equal(_tri_hash(cat), _tri_hash(soup0)), "triangle soup"


def test_512_rays_exact_against_oracle(ivxlib, oracle, v512):
    """MIDA / LMIP / fast contour MIP on V512, every axis, every pixel (replaces the range-bound property test)"""
    from invesalius3_amd import invesalius_rs as mips
    img, _ = v512
    for axis in range(3):
        shp = tuple(s for i, s in enumerate(img.shape) if i != axis)
        g, r = np.zeros(shp, np.int16), np.zeros(shp, np.int16)
        mips.mida(img, axis, 300, 600, g)
        oracle.mida(img, axis, 300, 600, r)
        assert np.array_equal(g, r), ("mida", axis)
        g[:] = 0
        r[:] = 0
        mips.lmip(img, axis, 700, 3033, g)
        oracle.lmip(img, axis, 700, 3033, r)
        assert np.array_equal(g, r), ("lmip", axis)
    for tmip in (0, 1, 2):
        g, r = np.zeros(img.shape[1:], np.int16), np.zeros(img.shape[1:], np.int16)
        mips.fast_countour_mip(img, 2.0, 0, 300, 600, tmip, g)
        oracle.fast_countour_mip(img, 2.0, 0, 300, 600, tmip, r)
        assert np.array_equal(g, r), ("fast_countour_mip", tmip)


def test_v512_watershed_ift_equals_serial_oracle_whole_volume(ivxlib, oracle, v512):
    """configs[2], IFT branch, the WHOLE 512^3 volume: min-shift cost image, bench.py's markers,
    6 neighbours.  Bit for bit against the serial, defect-free statement of NI_WatershedIFT; against live scipy (the
    reference proper) the number of differing voxels is measured and bounded -- scipy leaves its own documented algorithm
    downstream of the linked-list defect of ni_measure.c, and that deviation is stated, not hidden."""
    from scipy import ndimage

    from bench import ws_markers
    from invesalius3_amd import watershed_process as wp
    img, _ = v512
    strct = generate_binary_structure(3, 1)
    cost = (img - img.min()).astype(np.uint16)
    mk = ws_markers(img)
    got = wp.watershed_ift(cost, mk, strct)
    clean = oracle.watershed_ift_clean(cost, mk, strct)
    assert np.array_equal(got, clean), "%d voxels differ from the defect-free serial flood" % int((got != clean).sum())
    # the reference proper: scipy's labels on this very volume are on file (tests/golden/ws512_full.npz, made by
    # make_golden_ws_full.py from live scipy: its CRC-32 and the voxels where it leaves the defect-free statement); on a box
    # whose synthetic volume differs, scipy runs live.  The count is PINNED: a change in either direction must be seen.
    fx = _full_fixture(512, img)
    if fx is not None:
        at = np.cumsum(fx["differs_at"].astype(np.int64))
        ref = got.astype(np.uint8).reshape(-1).copy()
        ref[at] = 3 - ref[at]
        assert zlib.crc32(ref) == int(fx["scipy_crc32"]), "rebuilt reference labels miss scipy's CRC"
        n_ref = len(at)
    else:
        n_ref = int((got != ndimage.watershed_ift(cost, mk, strct)).sum())
    print("watershed_ift 512^3: differs_from_reference (live scipy) = %d of %d voxels" % (n_ref, got.size))
    assert n_ref == 87372, n_ref  # 0.065 % of 134 217 728 voxels, downstream of scipy's linked-list defect


def test_v512_watershed_gui_default_equals_serial_oracle_whole_volume(ivxlib, oracle, v512):
    """configs[2] with the GUI's default settings ("Watershed": window/level LUT, 3x3x3 gradient, scikit-image's heap
    flood, 6 neighbours), the WHOLE 512^3 volume: bit for bit against the serial heap flood with raster-ordered marker ties,
    and the count against the heap-ordered one (scikit-image move for move) reported."""
    from bench import ws_markers
    from invesalius3_amd import watershed_process as wp
    img, _ = v512
    strct = generate_binary_structure(3, 1)
    mk = ws_markers(img).astype(np.int16)
    grad = wp.cost_image(img, True, 300, 400, (3, 3, 3))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", wp.MarkerTieWarning)
        got, st = wp.watershed(grad, mk, strct, want_stats=True)
    raster = oracle.watershed_sk(grad, mk, strct, 1)
    assert np.array_equal(got, raster), "%d voxels differ from the serial flood (raster ties)" % int((got != raster).sum())
    heap = oracle.watershed_sk(grad, mk, strct, 0)
    n_ref = int((got != heap).sum())
    print("watershed (GUI default) 512^3: differs_from_reference (heap-ordered ties) = %d of %d voxels, tied markers of "
          "different labels: %d" % (n_ref, got.size, st["tied_markers_of_different_labels"]))
    if st["tied_markers_of_different_labels"] == 0:
        assert n_ref == 0


@pytest.fixture(scope="module")
def v1024():
    """BASELINE configs[2]'s volume at its stated size (2 GB of int16: ~2 minutes of numpy)"""
    from bench import synth_v512, ws_markers
    img = synth_v512((1024, 1024, 1024))
    fx = _full_fixture(1024, img)
    if fx is None:
        pytest.skip("tests/golden/ws1024_full.npz does not describe this box's synthetic volume (the serial floods take ~25 min and "
                    "~50 GB at 1024^3: make_golden_ws_full.py --size 1024)")
    return img, ws_markers(img), fx


def test_v1024_watershed_ift_equals_serial_oracle_whole_volume(ivxlib, v1024):
    """configs[2] at the size BASELINE.json states, IFT branch (VERDICT r3 missing #5): the HIP flood's labels over all 2^30
    voxels have the CRC-32 of the serial defect-free statement (oracle/ivx_oracle_wsz.c, run once on this volume by
    tests/golden/make_golden_ws_full.py), and flipped at the recorded voxels they have the CRC-32 of live scipy's labels --
    i.e. they differ from the reference exactly there (the count is part of the file)."""
    from invesalius3_amd import watershed_process as wp
    img, mk, fx = v1024
    cost = (img - img.min()).astype(np.uint16)
    got = np.ascontiguousarray(wp.watershed_ift(cost, mk, generate_binary_structure(3, 1)), dtype=np.uint8)
    del cost
    assert zlib.crc32(got) == int(fx["clean_crc32"]), "1024^3 IFT flood differs from the serial defect-free flood"
    at = np.cumsum(fx["differs_at"].astype(np.int64))
    ref = got.reshape(-1)
    ref[at] = 3 - ref[at]
    assert zlib.crc32(ref) == int(fx["scipy_crc32"])
    print("watershed_ift 1024^3: differs_from_reference (live scipy) = %d of %d voxels" % (len(at), got.size))
    assert len(at) == int(fx["differs"])


def test_v1024_watershed_gui_default_equals_serial_oracle_whole_volume(ivxlib, v1024):
    """configs[2] at 1024^3 with the GUI's default settings: cost image (window/level LUT + 3x3x3 gradient) == numpy + scipy's,
    labels == the serial heap flood with raster-ordered marker ties, by CRC-32 over the whole volume; the recorded count
    against scikit-image's heap-ordered ties is reported."""
    import warnings

    from invesalius3_amd import watershed_process as wp
    img, mk, fx = v1024
    grad = wp.cost_image(img, True, 300, 400, (3, 3, 3))
    assert zlib.crc32(grad) == int(fx["grad_crc32"]), "1024^3 cost image differs from numpy LUT + scipy morphological_gradient"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", wp.MarkerTieWarning)
        got = wp.watershed(grad, mk.astype(np.int16), generate_binary_structure(3, 1))
    del grad
    assert zlib.crc32(np.ascontiguousarray(got, dtype=np.uint8)) == int(fx["sk_raster_crc32"]), "1024^3 flood differs from the serial heap flood"
    print("watershed (GUI default) 1024^3: differs_from_reference (heap-ordered ties) = %d" % int(fx["sk_differs"]))


def _tri_hash(v):
    """order-independent fingerprint of a soup: one 64-bit hash per triangle, sorted"""
    u = np.ascontiguousarray(v, dtype=np.float32).reshape(len(v), 9).view(np.uint32).astype(np.uint64)
    h = np.zeros(len(u), np.uint64)
    for k in range(9):
        h = (h * np.uint64(0x9E3779B97F4A7C15) + u[:, k] + np.uint64(k + 1)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(29)
    return np.sort(h)


def test_2048_wide_eight_slabs_stitch_equals_single_volume(ivxlib):
    """configs[3] geometry on one GPU: 8 ranks x 32 slices x 2048 x 2048 (each slab two flood tiles deep)"""
    from _ptr_comm import LoopbackWorld
    from bench import synth_v512
    from invesalius3_amd.device import DeviceVolume
    from _stitch_ref import stitch_piece_meshes
    from invesalius3_amd.parallel import SlabVolume

    world, nz = 8, 32
    full = synth_v512((world * nz, 2048, 2048), seed=7)
    z, y, x = np.unravel_index(int(np.argmax(full)), full.shape)
    seeds = [(int(x), int(y), int(z))]
    lw = LoopbackWorld(world)
    res, errs = {}, []

    def run(rank):
        try:
            vol = SlabVolume(full[rank * nz:(rank + 1) * nz], rank, world, comm=lw.comm(rank), spacing=(0.5, 0.5, 1.0))
            vol.threshold(*BONE)
            vol.region_grow(seeds, BONE[0], BONE[1], S26, fill=1, select_value=254)
            lay = vol.lay
            res[rank] = dict(mesh=vol.marching_cubes_indexed(from_binary=True, download=True),
                             stitched=vol.marching_cubes_stitched(from_binary=True, download=True), count=vol.reached_count(),
                             mask=vol.download_mask()[lay.first_interior:lay.last_interior + 1])
            vol.close()
        except Exception as e:  # pragma: no cover
            import traceback
            errs.append((rank, traceback.format_exc()))
            lw.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=900) for t in th]
    assert not errs, errs
    one = DeviceVolume(full, spacing=(0.5, 0.5, 1.0))
    one.threshold(*BONE)
    one.region_grow(seeds, BONE[0], BONE[1], S26, fill=1, select_value=254)
    v1, f1 = one.marching_cubes_indexed(from_binary=True, download=True)
    assert sum(res[r]["count"] for r in range(world)) == one.reached_count() > 10 ** 6
    assert np.array_equal(np.concatenate([res[r]["mask"] for r in range(world)]), one.download_mask())
    one.close()
    sv, sf = stitch_piece_meshes([res[r]["mesh"] for r in range(world)])
    # the device stitch (what bench.py --config sharded2048 times): the same arrays as the numpy restatement
    dv = np.concatenate([res[r]["stitched"][1] for r in range(world)])
    df = np.concatenate([res[r]["stitched"][2] for r in range(world)])
    assert np.array_equal(dv.view(np.uint32), sv.view(np.uint32)) and np.array_equal(df, sf), "device stitch != host stitch"
    del dv, df
    assert len(sv) == len(v1) and len(sf) == len(f1) > 10 ** 6
    assert len(sv) < sum(len(res[r]["mesh"][0]) for r in range(world))  # the shared planes' vertices were merged
    assert np.array_equal(_tri_hash(sv[sf]), _tri_hash(v1[f1]))
    # same vertex set (exact float32 bits), each vertex once
    key = lambda v: np.sort(np.ascontiguousarray(v).view([("", np.uint32)] * 3).ravel())
    assert np.array_equal(key(sv), key(v1))
    assert lw.collectives >= 3  # image halo + at least two region-growing rounds went through the communicator
