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


def test_do_watershed_says_where_it_leaves_the_reference_and_only_after_the_queue_is_signalled(tmp_path, monkeypatch):
    """the IFT branch of do_watershed computes the defect-free statement of scipy's flood, which differs
    from live scipy (the reference, watershed_process.py:44-46,54-57) on realistic volumes.  The hook must SAY so where the
    user is: `last_stats` carries reference / exact / note, an `IftDeviationWarning` is raised once per process, and -- like
    the scikit-image branch's `MarkerTieWarning` -- only after the labels are in the memmap, flushed, and `q` has its
    signal (with -W error the caller waiting on q, styles.py:2116-2134, must not hang).  The device call is a stand-in that
    writes known labels; everything else is the shipped host code."""
    import ctypes
    import warnings

    import numpy as np
    import pytest
    from invesalius3_amd import watershed_process as wp

    shape = (3, 4, 5)
    calls = []

    class Lib:
        def ivx_do_watershed_into(self, img, ishape, istr, mk_code, mk, mk_str, mdt, strct, alg, gs, uw, ww, wl, out, ostr, stats):
            calls.append((mk_code, mdt, alg, list(ostr)))
            n = int(np.prod(shape))
            ctypes.memmove(out.value, (np.arange(n) % 3).astype(np.uint8).ctypes.data, n)
            stats[6] = 2 if alg == 1 else 0
            return 0

    monkeypatch.setattr(wp.L, "lib", lambda: Lib())
    monkeypatch.setattr(wp, "_ift_warned", False)
    tfile = str(tmp_path / "m.dat")
    np.zeros(shape, np.uint8).tofile(tfile)
    img = np.zeros(shape, np.int16)
    mk = np.zeros(shape, np.int16)
    st = np.ones((3, 3, 3), bool)
    events = []

    class Q:
        def put(self, v):
            # the labels must already be on disk when the signal goes out
            events.append(("put", v, bytes(np.fromfile(tfile, np.uint8)[:4])))

    with warnings.catch_warnings():
        warnings.simplefilter("error")  # -W error: the warning becomes an exception ...
        with pytest.raises(wp.IftDeviationWarning):
            wp.do_watershed(img, mk, tfile, shape, st, "Watershed IFT", (3, 3, 3), True, 300, 400, Q())
    assert events == [("put", 1, bytes([0, 1, 2, 0]))]  # ... and it came after the write and the signal
    ls = wp.do_watershed.last_stats
    assert ls["algorithm"] == "Watershed IFT" and ls["reference"] == "scipy.ndimage.watershed_ift" and ls["exact"] is False
    assert "87 372" in ls["note"] and "206 606" in ls["note"]
    # int16 markers with ww/wl: passed in their own dtype, cast to int16 on the device; labels straight into the memmap (dense strides)
    assert calls[-1] == (wp.L.I16, wp.L.I16, 0, [20, 5, 1])
    # once per process: the second call is silent, still describes itself
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        wp.do_watershed(img, mk.astype(np.int32), tfile, shape, st, "Watershed IFT", (3, 3, 3), False, 0, 0, Q())
    assert calls[-1][:3] == (wp.L.I32, wp.L.I8, 0) and len(events) == 2  # min-shift IFT branch: int8 (watershed_process.py:57)
    assert wp.do_watershed.last_stats["exact"] is False
    # the scikit-image branch keeps its own warning, same placement
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with pytest.raises(wp.MarkerTieWarning):
            wp.do_watershed(img, mk, tfile, shape, st, "Watershed", (3, 3, 3), True, 300, 400, Q())
    assert len(events) == 3 and wp.do_watershed.last_stats["tied_markers_of_different_labels"] == 2
    assert wp.do_watershed.last_stats["reference"] == "skimage.segmentation.watershed" and wp.do_watershed.last_stats["exact"] is False
    # float markers (not castable on the device) take the reference's host cast
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wp.do_watershed(img, mk.astype(np.float64), tfile, shape, st, "Watershed", (3, 3, 3), True, 300, 400, None)
    assert calls[-1][:2] == (wp.L.I16, wp.L.I16)
