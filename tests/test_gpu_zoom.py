"""resize_image_array = scipy.ndimage.zoom(image, factor, image.dtype, order=2) on the GPU (csrc/k_zoom.hip) against live
scipy -- the function the reference calls (imagedata_utils.py:121-130) -- and the reference-signature
create_surface_piece / .vtp round trip built on it (surface.py:1350-1410)."""
import os

import numpy as np
import pytest
from scipy import ndimage

from conftest import synth_volume

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,factor", [((7, 9, 13), 0.5), ((20, 31, 17), 1 / 3.0), ((40, 37, 50), 0.5), ((33, 64, 65), 0.5),
                                          ((1, 12, 12), 0.5), ((64, 96, 96), 1 / 3.0)])
def test_zoom_order2_equals_scipy(ivxlib, shape, factor):
    from invesalius3_amd import surface_process as sp
    img = synth_volume(shape, seed=9)
    exp = ndimage.zoom(img, factor, img.dtype, order=2)
    got = sp.resize_image_array(img, factor)
    assert got.dtype == np.int16 and got.shape == exp.shape and np.array_equal(got, exp)
    mask = np.where(img > 200, 255, 0).astype(np.uint8)
    mask[img > 900] = 254
    exp = ndimage.zoom(mask, factor, mask.dtype, order=2)
    got = sp.resize_image_array(mask, factor)
    assert got.dtype == np.uint8 and np.array_equal(got, exp)


def test_zoom_medium_quality_volume_and_memmap(ivxlib):
    """the Medium preset on a 128 x 160 x 160 volume and its (d+1, h+1, w+1) mask, as AddNewActor does it"""
    from invesalius3_amd import surface_process as sp
    img = synth_volume((128, 160, 160), seed=10)
    mask = np.zeros((129, 161, 161), np.uint8)
    mask[1:, 1:, 1:] = np.where(img >= 226, 255, 0)
    for arr in (img, mask):
        got = sp.resize_image_array(arr, 0.5, as_mmap=True)
        assert isinstance(got, np.memmap)
        assert np.array_equal(np.asarray(got), ndimage.zoom(arr, 0.5, arr.dtype, order=2))
        os.remove(got.filename)


def test_create_surface_piece_reference_signature(ivxlib, oracle, tmp_path):
    """called with exactly the argument list surface.py:1381-1410 passes: memmap file names in, .vtp name out; the pieces
    joined (append + clean) equal the whole-volume indexed surface"""
    from invesalius3_amd import surface_process as sp
    img = synth_volume((45, 48, 64), seed=11)
    mask = np.zeros((46, 49, 65), np.uint8)
    mask[1:, 1:, 1:] = np.where(img >= 226, 255, 0)
    f_img, f_mask = str(tmp_path / "img.dat"), str(tmp_path / "mask.dat")
    np.memmap(f_img, dtype=np.int16, mode="w+", shape=img.shape)[:] = img
    np.memmap(f_mask, dtype=np.uint8, mode="w+", shape=mask.shape)[:] = mask
    spacing = (0.5, 0.5, 2.0)
    for from_binary, algorithm in ((True, "ca_smoothing"), (False, "Default")):
        names = []
        n_pieces = int(round(img.shape[0] / 20 + 0.5, 0))
        for i in range(n_pieces):
            roi = slice(i * 20, i * 20 + 21)
            name = sp.create_surface_piece(f_img, img.shape, img.dtype, f_mask, mask.shape, mask.dtype, roi, spacing, "CONTOUR",
                                           226, 3071, 0.0, 0.0, 0, "en", True, from_binary, algorithm, 0, True)
            assert name.endswith("_%d_%d.vtp" % (roi.start, roi.stop)) and os.path.exists(name)
            v, f = sp.read_vtp(name)
            want = oracle.create_surface_piece(img, mask, roi, spacing, 226, 3071, from_binary)
            assert np.array_equal(v[f], want)
            names.append(name)
        verts, faces, m = sp.join_surface_pieces(names[::-1])      # any order in: sorted by the roi in the name
        wv, wf, wm = sp.join_process_volume(img, mask, spacing, 226, 3071, from_binary)
        key = lambda t: np.sort(np.ascontiguousarray(t).reshape(len(t), -1).view([("", np.float32)] * 9), axis=0)
        assert len(verts) == len(wv) and np.array_equal(key(verts[faces]), key(wv[wf]))
        assert m["area"] == pytest.approx(wm["area"], rel=1e-12) and m["volume"] == pytest.approx(wm["volume"], rel=1e-9)
        # the reference's own join step, with its argument list (surface.py:1440-1452) and its queue messages
        import queue
        q = queue.Queue()
        opts = {"angle": 0.7, "max distance": 3.0, "min weight": 0.5, "steps": 3}
        full, meas = sp.join_process_surface(names, "Default", 0, 0.0, 0.0, True, True, opts, q)
        assert full.endswith("_full.vtp")
        fv, ff = sp.read_vtp(full)
        lv, lf, _ = sp.keep_largest(wv, wf)
        # (the file holds the points split at feature edges, with normals -- surface_process.point_normals: the same
        # triangles in space, more points than the merged surface the measures were taken on)
        assert len(fv) >= len(lv) and np.array_equal(fv[ff], lv[lf])
        vol, area = sp.mass_properties(lv, lf)
        assert meas["area"] == pytest.approx(area, rel=1e-12) and meas["volume"] == pytest.approx(vol, rel=1e-9)
        msgs = []
        while not q.empty():
            msgs.append(q.get())
        assert msgs == ["Joining surfaces ...", "Cleaning surface ...", "Decimating ...", "Finding the largest ...",
                        "Filling holes ...", "Calculating area and volume ..."]
        os.remove(full)
        full, meas_s = sp.join_process_surface(names, "ca_smoothing", 0, 0.0, 0.0, False, False, opts, None)
        sv, sf = sp.read_vtp(full)
        from invesalius3_amd import invesalius_rs as rs
        mesh = rs.Mesh.from_indexed(np.array(verts, np.float32), faces)  # (a copy: the smoothing works in place)
        rs.ca_smoothing(mesh, 0.7, 3.0, 0.5, 3)
        smoothed = np.asarray(mesh.vertices, np.float32)
        assert np.array_equal(sv[sf], smoothed[faces]) and not np.array_equal(smoothed, verts)
        os.remove(full)
        for n in names:
            os.remove(n)
    # the "InVesalius 3.b2" value rewrite (surface_process.py:128-146), reachable only with from_binary=False
    mask2 = mask.copy()
    mask2[1:, 1:, 1:][img > 1200] = 254
    mask2[1:, 1:, 1:][(img > 100) & (img < 226)] = 1
    np.memmap(f_mask, dtype=np.uint8, mode="r+", shape=mask.shape)[:] = mask2
    roi = slice(0, 21)
    name = sp.create_surface_piece(f_img, img.shape, img.dtype, f_mask, mask.shape, mask.dtype, roi, spacing, "CONTOUR", 226, 3071,
                                   0.0, 0.0, 0, "en", True, False, "InVesalius 3.b2", 0, False)
    v, f = sp.read_vtp(name)
    a = img[roi].copy()
    am = mask2[1:22, 1:, 1:]
    a[am == 1] = np.array(int(a.min()) - 1).astype(np.int16)
    a[am == 254] = (226 + 3071) / 2.0
    want = oracle.marching_cubes(a, spacing, [226.0, 3071.0], 0, False, False, False, float(np.iinfo(np.int16).min), 0)
    assert len(f) > 0 and np.array_equal(v[f], want)
    os.remove(name)
