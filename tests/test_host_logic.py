"""CPU: host-side logic of the package that needs no device (argument checking, bounds, STL sink, layout helpers)."""
import numpy as np
import pytest


def test_integer_threshold_bounds():
    from invesalius3_amd.slice_ import _int_bounds
    assert _int_bounds((226, 3071)) == (226, 3071)
    assert _int_bounds((225.5, 3071.9)) == (226, 3071)  # v >= 225.5  <=>  v >= 226 for integer voxels
    assert _int_bounds((-1e12, 1e12)) == (-(2 ** 31), 2 ** 31 - 1)


def test_argument_errors_do_not_need_a_device():
    from invesalius3_amd import invesalius_rs as rs, slice_, surface_process as sp
    img = np.zeros((2, 3, 4), np.int16)
    with pytest.raises(TypeError):
        slice_.do_threshold_to_all_slices(np.zeros((3, 4, 5), np.int16), img, (0, 1))
    with pytest.raises(ValueError):
        slice_.do_threshold_to_all_slices(np.zeros((2, 3, 4), np.uint8), img, (0, 1))
    with pytest.raises(TypeError):
        rs.floodfill_threshold(img.astype(np.float32), [(0, 0, 0)], 0, 1, 1, np.ones((3, 3, 3)), np.zeros((2, 3, 4), np.uint8))
    with pytest.raises(TypeError):
        rs.floodfill_threshold(img, [(0, 0, 0)], 0, 1, 1, np.ones((3, 3, 3)), np.zeros((2, 3, 4), np.int16))
    with pytest.raises(TypeError):
        rs.mida(img, 0, 1, 1, np.zeros((3, 4), np.uint8))
    with pytest.raises(OverflowError):
        rs.mida(img, 0, 70000, 1, np.zeros((3, 4), np.int16))
    with pytest.raises(ValueError):
        sp.marching_cubes(np.zeros((2, 2, 2), np.uint8), (1, 1, 1), [1, 2, 3])
    with pytest.raises(TypeError):
        sp.marching_cubes(np.zeros((2, 2, 2), np.float32), (1, 1, 1), [1])


def test_stl_sink_layout(tmp_path):
    """vtkSTLWriter binary layout (surface.py:1827-1829): 80 B header, u32 count, 50 B per triangle"""
    from invesalius3_amd import surface_process as sp
    tris = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 1], [0, 1, 1], [1, 0, 1]]], np.float32)
    p = tmp_path / "s.stl"
    sp.write_stl_binary(str(p), tris)
    raw = p.read_bytes()
    assert len(raw) == 80 + 4 + 2 * 50
    assert raw[:40] == b"Visualization Toolkit generated SLA File"
    assert np.frombuffer(raw[80:84], "<u4")[0] == 2
    rec = np.frombuffer(raw[84:], dtype=[("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    assert np.array_equal(rec["v"], tris)
    np.testing.assert_allclose(rec["n"], [[0, 0, 1], [0, 0, -1]])


def test_structures_accepted_by_the_union_find_engine():
    """host-side mirror of ccl_supported(): symmetric strct with both x neighbours in the centre row"""
    from scipy.ndimage import generate_binary_structure

    def bits(s):
        s = np.asarray(s, np.uint8)
        return sum(1 << k for k, v in enumerate(s.ravel()) if v)

    def supported(b):
        b |= 1 << 13
        return all(((b >> k) & 1) == ((b >> (26 - k)) & 1) for k in range(27)) and (b >> 12) & 1 and (b >> 14) & 1
    for conn in (1, 2, 3):
        assert supported(bits(generate_binary_structure(3, conn)))
    one_way = np.zeros((3, 3, 3), np.uint8)
    one_way[1, 1, 2] = 1
    assert not supported(bits(one_way))


def test_mesh_oracle_anchors(oracle):
    """the mesh oracle (parity unpinned vs VTK) against analytic shapes and scipy's connected components"""
    # unit cube, outward triangles
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7],
                  [2, 7, 6], [3, 0, 4], [3, 4, 7]], np.int32)
    m = oracle.mesh_mass_properties(v * np.float32(2.0), f)
    assert m[0] == pytest.approx(8.0, rel=1e-12) and m[1] == pytest.approx(24.0, rel=1e-12)
    assert m[5] + m[6] + m[7] == pytest.approx(1.0)
    # two cubes + a lone triangle: the first cube wins the tie, vertices are compacted
    v2 = np.concatenate([v, v + np.float32(5.0), np.zeros((3, 3), np.float32)])
    f2 = np.concatenate([f + 8, f, [[16, 17, 18]]]).astype(np.int32)
    kv, kf, nreg = oracle.mesh_keep_largest(v2, f2)
    assert nreg == 3 and np.array_equal(kf, f) and np.array_equal(kv, v + np.float32(5.0))
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    e = np.concatenate([f2[:, [0, 1]], f2[:, [0, 2]]])
    g = coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(len(v2), len(v2)))
    assert connected_components(g, directed=False)[0] == 3


def test_stitch_piece_meshes_merges_the_shared_planes(oracle):
    """cross-slab stitch: rank pieces (indexed) -> one mesh with every vertex once, same triangles as the whole volume"""
    from conftest import synth_volume
    from invesalius3_amd import parallel as par
    img = synth_volume((30, 24, 40), seed=3)
    mask = np.where(img > 100, 255, 0).astype(np.uint8)

    def index(soup):
        u, inv = np.unique(soup.reshape(-1, 3), axis=0, return_inverse=True)
        return u.astype(np.float32), inv.reshape(-1, 3).astype(np.int32)

    world, nz, pieces = 3, 10, []
    for r in range(world):
        lay = par.slab_layout(r, world, nz)
        a = par.slab_mc_args(lay)
        lo = lay.z_global0 - lay.hb
        piece = mask[lo: lo + lay.local_dz][a["z0"]: a["z1"]]
        pieces.append(index(oracle.marching_cubes(piece, (0.5, 0.5, 2.0), [127.0], a["roi_start"], True, a["pad_bottom"],
                                                  a["pad_top"], 0.0, int(a["pad_bottom"]))))
    from _stitch_ref import stitch_piece_meshes  # (the numpy restatement the device stitch is checked against)
    v, f = stitch_piece_meshes(pieces)
    whole = oracle.marching_cubes(mask, (0.5, 0.5, 2.0), [127.0], 0, True, True, True, 0.0, 1)
    assert len(v) == len(np.unique(whole.reshape(-1, 3), axis=0)) == len(np.unique(v, axis=0))
    assert len(v) < sum(len(p[0]) for p in pieces)  # something was merged
    key = lambda t: np.sort(t.reshape(len(t), -1).view([("", np.float32)] * 9), axis=0)
    assert f.dtype == np.int32 and np.array_equal(key(v[f]), key(whole))
    # degenerate inputs
    ev, ef = stitch_piece_meshes([])
    assert ev.shape == (0, 3) and ef.shape == (0, 3)
    one_v, one_f = stitch_piece_meshes([pieces[0]])
    assert np.array_equal(one_v, pieces[0][0]) and np.array_equal(one_f, pieces[0][1])


def test_structure_helper_is_scipys_generate_binary_structure():
    from scipy.ndimage import generate_binary_structure
    from invesalius3_amd.mask import CON2D, CON3D, _structure
    for conn, c in CON3D.items():
        assert np.array_equal(_structure(3, c), generate_binary_structure(3, c)) and _structure(3, c).sum() == conn + 1
    for conn, c in CON2D.items():
        assert np.array_equal(_structure(2, c), generate_binary_structure(2, c)) and _structure(2, c).sum() == conn + 1


def test_headless_argument_parsing():
    from invesalius3_amd import headless
    with pytest.raises(SystemExit):
        headless.main(["case.inv3", "--seed", "1", "2"])          # seeds come in triples
    with pytest.raises(SystemExit):
        headless.main(["case.inv3", "--threshold", "1", "2", "--mask", "1"])  # either a new threshold or a saved mask


def test_slab_region_grow_counts_its_collectives_and_host_reads():
    """slab_region_grow against scripted stand-ins: a flood that gains bits in k rounds costs k + 1 collectives (the vote
    of round k travels with round k + 1) and k + 1 host reads; a rank floods only after a round in which its halo gained
    something; end ranks pass no pointers for the neighbour they do not have."""
    from invesalius3_amd import parallel as par

    class Backend:
        def __init__(self, gains):
            self.gains, self.floods, self.ors, self.reads, self.staged = list(gains), 0, 0, 0, 0
            self.vote, self.changed = 0, 1  # the seeds count as "something happened"

        def flood_run(self):
            self.floods += 1

        def stage_vote(self):
            self.staged += 1
            self.vote = self.changed

        def round_ptrs(self):
            return ("down", "rdown", "up", "rup", 64, self, "stream")

        def or_planes(self):
            self.ors += 1
            self.changed = self.gains.pop(0) if self.gains else 0

        def read_votes(self):
            self.reads += 1
            return self.vote, self.changed

    class Comm:
        """two ranks' worth of votes: `other` is what the peer contributes to the same rounds' all-reduce"""
        def __init__(self, other):
            self.other, self.calls, self.args = list(other), 0, []

        def exchange_vote(self, to_down, from_down, to_up, from_up, nbytes, vote, nvote, stream):
            self.calls += 1
            self.args.append((to_down, from_down, to_up, from_up, nbytes, nvote, stream))
            vote.vote += self.other.pop(0) if self.other else 0

    lay = par.slab_layout(1, 3, 8)  # a middle rank: both neighbours exist
    # this rank gains in rounds 1 and 2, the peer only in round 1: rounds 1, 2 productive, round 3 gains nothing, round 4
    # learns that -> 4 collectives, 4 host reads, 3 floods (initial + after the two productive rounds)
    be, comm = Backend([5, 2, 0]), Comm([1, 3, 0, 0])  # peer's staged votes: seeds, then its gains
    rounds = par.slab_region_grow(be, comm, lay)
    assert (rounds, comm.calls, be.floods, be.ors, be.reads, be.staged) == (4, 4, 3, 4, 4, 4)
    assert comm.args[0] == ("down", "rdown", "up", "rup", 64, 1, "stream")
    # nothing ever gained: one round to exchange, a second one to learn that nobody gained -> 2 collectives, 1 flood
    be, comm = Backend([0]), Comm([1, 0])
    assert par.slab_region_grow(be, comm, lay) == 2 and (comm.calls, be.floods, be.reads) == (2, 1, 2)
    # end ranks: no pointers towards the missing neighbour
    be, comm = Backend([0]), Comm([1, 0])
    par.slab_region_grow(be, comm, par.slab_layout(0, 3, 8))
    assert comm.args[0][:4] == (None, None, "up", "rup")
    be, comm = Backend([0]), Comm([1, 0])
    par.slab_region_grow(be, comm, par.slab_layout(2, 3, 8))
    assert comm.args[0][:4] == ("down", "rdown", None, None)


def test_get_image_slice_plain_slices_and_the_missing_lmip_export():
    """slice_.get_image_slice without a projection is pure slicing (no device needed): one slice along each orientation,
    whatever slab thickness is asked for (slice_.py:847-848); the LMIP type dies like the reference's missing export."""
    import numpy as np
    import pytest
    from invesalius3_amd import slice_ as sl
    a = np.arange(4 * 5 * 6, dtype=np.int16).reshape(4, 5, 6)
    assert np.array_equal(sl.get_image_slice(a, "AXIAL", 2, 3), a[2])
    assert np.array_equal(sl.get_image_slice(a, "CORONAL", 1, 4, inverted=True), a[:, 1, :])
    assert np.array_equal(sl.get_image_slice(a, "SAGITAL", 5), a[:, :, 5])
    with pytest.raises(AttributeError):
        sl.get_image_slice(a, "AXIAL", 0, 2, False, 1.0, sl.PROJECTION_LMIP, 100)
    with pytest.raises(KeyError):
        sl.get_image_slice(a, "OBLIQUE", 0)


def _sphere(levels=3, inside_out=False):
    """an octahedron subdivided `levels` times onto the unit sphere: closed, outward wound, all dihedral angles small"""
    v = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    v = [np.array(p, np.float64) for p in v]
    for _ in range(levels):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        f = nf
    f = np.array(f, np.int32)
    return np.array(v, np.float32), (f[:, ::-1].copy() if inside_out else f)


def test_point_normals_split_at_feature_edges_and_point_outwards():
    """the vtkPolyDataNormals step of join_process_surface (surface_process.py:420-435), pinned by properties (VTK absent)"""
    from invesalius3_amd import surface_process as sp
    cube_v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.float32)
    cube_f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6],
                       [3, 0, 4], [3, 4, 7]], np.int32)
    v, f, pn, cn = sp.point_normals(cube_v, cube_f)
    assert len(v) == 24 and len(f) == 12 and pn.dtype == np.float32        # every corner is three points, one per face
    assert np.allclose(np.linalg.norm(pn, axis=1), 1.0) and np.allclose(np.linalg.norm(cn, axis=1), 1.0)
    assert np.array_equal(v[f].reshape(-1, 3), cube_v[cube_f].reshape(-1, 3))  # same triangles in space
    assert np.allclose(pn[f[:, 0]], cn) and np.allclose(pn[f[:, 2]], cn)       # a face's points carry the face normal
    centre = v[f].mean(axis=(0, 1))
    assert (np.einsum("ij,ij->i", cn, v[f].mean(axis=1) - centre) > 0).all()      # outwards
    vi, fi, pni, cni = sp.point_normals(cube_v, cube_f[:, ::-1])                  # wound inside out: turned around
    assert (np.einsum("ij,ij->i", cni, vi[fi].mean(axis=1) - centre) > 0).all()
    v0, f0, pn0, _ = sp.point_normals(cube_v, cube_f, splitting=False)
    assert len(v0) == 8 and np.array_equal(f0, cube_f)
    sv, sf = _sphere(3)
    v, f, pn, cn = sp.point_normals(sv, sf)
    assert len(v) == len(sv) and np.array_equal(f, sf)                         # smooth everywhere: nothing is split
    assert np.einsum("ij,ij->i", pn, sv).min() > 0.99                          # the sphere's normals are its points
    v, f, pn, cn = sp.point_normals(*_sphere(3, inside_out=True))
    assert np.einsum("ij,ij->i", pn, sv).min() > 0.99


def test_fill_holes_caps_the_rims_up_to_the_hole_size():
    """the vtkFillHolesFilter step (surface_process.py:396-416, hole size 300), pinned by properties"""
    from invesalius3_amd import surface_process as sp
    sv, sf = _sphere(3)
    assert len(sp.boundary_edges(sf)) == 0
    v, f, n = sp.fill_holes(sv, sf)
    assert n == 0 and v is not None and np.array_equal(f, sf)                  # closed: unchanged
    top = sv[sf].mean(axis=1)[:, 2] > 0.8
    open_f = sf[~top]                                                          # a cap cut off
    rim = sp.boundary_edges(open_f)
    assert len(rim) > 0
    v, f, n = sp.fill_holes(sv, open_f)
    assert n == 1 and len(v) == len(sv) + 1 and len(f) == len(open_f) + len(rim)
    assert len(sp.boundary_edges(f)) == 0                                      # closed again, every edge twice
    assert np.array_equal(f[:len(open_f)], open_f)                             # the new triangles follow the old ones
    p = v.astype(np.float64)
    signed = np.einsum("ij,ij->i", p[f[:, 0]], np.cross(p[f[:, 1]], p[f[:, 2]])).sum() / 6
    full = np.einsum("ij,ij->i", sv[sf[:, 0]].astype(np.float64), np.cross(sv[sf[:, 1]], sv[sf[:, 2]])).sum() / 6
    assert 0.8 * full < signed < full                                          # the flat cap cuts a little of the ball off
    v2, f2, n2 = sp.fill_holes(sv * 1000.0, open_f)                            # the same rim, 1000x larger: above the hole size
    assert n2 == 0 and len(f2) == len(open_f)
    bottom = sv[sf].mean(axis=1)[:, 2] < -0.8
    v3, f3, n3 = sp.fill_holes(sv, sf[~top & ~bottom])                         # two rims: two caps
    assert n3 == 2 and len(sp.boundary_edges(f3)) == 0


def test_bench_launcher_gives_every_rank_the_same_nonce_and_the_rendezvous_completes(tmp_path):
    """`python bench.py --gpus N` outside a launcher: the N rank environments must share ONE
    IVX_COMM_NONCE, otherwise ranks > 0 never accept rank 0's id file.  Runs the rendezvous itself (comm.exchange_id, no
    device) in three fresh processes with exactly the environments bench.rank_envs() hands to its ranks; rank 2 arrives late."""
    import os
    import subprocess
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    idfile = str(tmp_path / "comm.id")
    envs = bench.rank_envs(3, idfile)
    assert len({e["IVX_COMM_NONCE"] for e in envs}) == 1 and [e["RANK"] for e in envs] == ["0", "1", "2"]
    code = ("import os, sys, time; sys.path.insert(0, %r); from invesalius3_amd import comm; r = int(os.environ['RANK']); "
            "time.sleep(1.0 if r == 2 else 0.0); "
            "print(comm.exchange_id(r, comm.rendezvous_file(), lambda: bytes(range(128)), timeout_s=20.0).hex())"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    procs = [subprocess.Popen([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE) for e in envs]
    outs = [p.communicate(timeout=60)[0].decode().strip() for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0]
    assert outs == [bytes(range(128)).hex()] * 3
    # a file of ANOTHER launch (different nonce) under the same name is never accepted
    other = dict(bench.rank_envs(2, idfile)[1])
    code2 = code.replace("timeout_s=20.0", "timeout_s=0.5")
    p = subprocess.run([sys.executable, "-c", code2], env=other, capture_output=True)
    assert p.returncode != 0 and b"no RCCL id of this launch" in p.stderr


def test_fill_holes_at_a_pinch_point_closes_both_rims():
    """two holes that touch in ONE vertex: that vertex has two outgoing rim edges; every rim edge must be
    consumed exactly once and no boundary edge may remain after capping"""
    from invesalius3_amd import surface_process as sp
    n = 6
    ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    verts = np.stack([ii.ravel(), jj.ravel(), np.zeros(ii.size)], axis=1).astype(np.float32)
    vid = lambda i, j: i * (n + 1) + j
    faces = []
    for i in range(n):
        for j in range(n):
            if (i, j) in ((1, 1), (2, 2)):  # two missing cells sharing the corner (2, 2)
                continue
            faces += [[vid(i, j), vid(i + 1, j), vid(i + 1, j + 1)], [vid(i, j), vid(i + 1, j + 1), vid(i, j + 1)]]
    faces = np.asarray(faces, np.int32)
    be = sp.boundary_edges(faces)
    loops = sp.boundary_loops(be)
    assert sorted(len(l) for l in loops) == [4, 4, 4 * n]             # two square rims + the sheet's outer rim, each simple
    assert sum(len(l) for l in loops) == len(be)                       # every rim edge consumed exactly once
    v, f, holes = sp.fill_holes(verts, faces, hole_size=1.0)           # (the outer rim is larger than the hole size)
    assert holes == 2 and len(v) == len(verts) + 2
    left = sp.boundary_edges(f)
    assert len(left) == 4 * n                                          # only the outer rim stays open
    assert not np.isin(left.ravel(), [vid(2, 2)]).any()
