"""Headless driver end to end: .inv3 in -> GPU threshold / region growing / indexed surface / largest region /
context-aware smoothing / mass properties -> STL + .inv3 out, checked stage by stage against the CPU oracle."""
import json
import struct

import numpy as np
import pytest

from conftest import synth_volume

pytestmark = pytest.mark.gpu


def _case(tmp_path, shape=(36, 40, 56)):
    from invesalius3_amd import project as prj
    img = synth_volume(shape, seed=31)
    p = prj.Project(name="Synth", spacing=(0.5, 0.5, 1.0), threshold_range=(int(img.min()), int(img.max())))
    p.matrix = img
    m = prj.new_mask(p, "saved", (150, 3071))
    m.matrix[1:, 1:, 1:] = np.where((img >= 150) & (img <= 3071), 255, 0)
    path = tmp_path / "synth.inv3"
    prj.save_inv3(path, p)
    return img, path


def _run(capsys, argv):
    from invesalius3_amd import headless
    assert headless.main([str(a) for a in argv]) == 0
    return json.loads(capsys.readouterr().out.strip().splitlines()[-1])


def test_threshold_seed_largest_smooth_stl_save(ivxlib, oracle, tmp_path, capsys):
    from invesalius3_amd import project as prj
    img, path = _case(tmp_path)
    lo, hi = 200, 3071
    cand = (img >= lo) & (img <= hi)
    z, y, x = (int(v[0]) for v in np.nonzero(cand))
    stl, saved = tmp_path / "out.stl", tmp_path / "out.inv3"
    res = _run(capsys, [path, "--threshold", lo, hi, "--seed", x, y, z, "--largest", "--smooth", "--steps", 4, "--stl", stl,
                        "--save", saved])
    # region growing: 26-connected component of the seed inside [lo, hi]
    grown = np.zeros(img.shape, np.uint8)
    oracle.floodfill_threshold(img, [(x, y, z)], lo, hi, 1, np.ones((3, 3, 3), np.uint8), grown)
    assert res["region_grow"]["voxels"] == int(grown.sum()) == res["mask_voxels"]
    mask = (grown * 255).astype(np.uint8)
    q = prj.open_inv3(saved)
    try:
        assert sorted(q.masks) == [0, 1] and q.masks[1].name == "GPU mask" and q.masks[1].threshold_range == (lo, hi)
        assert np.array_equal(q.masks[1].interior, mask) and (q.masks[1].matrix[1:, 0, 0] == 1).all()
        assert np.array_equal(q.matrix, img)
    finally:
        q.close()
    # surface chain on the oracle
    soup = oracle.marching_cubes(mask, (0.5, 0.5, 1.0), [127.0], 0, True, True, True, 0.0, 1)
    uniq, inv = np.unique(soup.reshape(-1, 3), axis=0, return_inverse=True)
    assert res["surface"] == {"vertices": len(uniq), "triangles": len(soup)}
    from invesalius3_amd import surface_process as sp
    verts, faces = sp.marching_cubes_indexed(mask, (0.5, 0.5, 1.0), [127.0], 0, True, True, True, 0.0, 1)
    assert np.array_equal(verts[faces], soup)
    v0, f0, nreg = oracle.mesh_keep_largest(verts, faces)
    assert res["largest"] == {"regions": nreg, "vertices": len(v0), "triangles": len(f0)}
    f4 = np.concatenate([np.full((len(f0), 1), 3), f0], axis=1).astype(np.int64)
    want = v0.copy()
    oracle.context_aware_smoothing(want, f4, oracle.mesh_face_normals(v0, f0), 0.7, 3.0, 0.5, 4)
    mass = oracle.mesh_mass_properties(want, f0)
    assert res["volume"] == pytest.approx(mass[0], rel=1e-10) and res["area"] == pytest.approx(mass[1], rel=1e-10)
    # the STL holds exactly the smoothed triangles
    raw = open(stl, "rb").read()
    n = struct.unpack("<I", raw[80:84])[0]
    rec = np.frombuffer(raw[84:], dtype=[("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    assert n == len(f0) == len(rec) and np.array_equal(rec["v"], want[f0])


def test_saved_mask_path(ivxlib, oracle, tmp_path, capsys):
    img, path = _case(tmp_path)
    res = _run(capsys, [path, "--mask", 0])
    mask = np.where((img >= 150) & (img <= 3071), 255, 0).astype(np.uint8)
    assert res["mask"]["name"] == "saved" and res["mask_voxels"] == int((mask > 0).sum())
    soup = oracle.marching_cubes(mask, (0.5, 0.5, 1.0), [127.0], 0, True, True, True, 0.0, 1)
    assert res["surface"]["triangles"] == len(soup)
    m = oracle.mesh_mass_properties(soup)
    assert res["volume"] == pytest.approx(m[0], rel=1e-9) and res["area"] == pytest.approx(m[1], rel=1e-9)
    with pytest.raises(KeyError):
        _run(capsys, [path, "--mask", 3])
