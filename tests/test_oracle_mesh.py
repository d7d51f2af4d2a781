"""CPU checks of the context-aware-smoothing oracle (C restatement of invesalius_rs/src/mesh.rs) against an
independent pure-Python transcription on small meshes, and of the quirks it carries."""
import numpy as np
import pytest


def _octa():
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [5, 5, 5]], np.float32)
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int64)
    return v, f


def _py_connectivity(f3, nv):
    adj = [[] for _ in range(nv)]
    for face in f3:
        for vi in face:
            for vj in face:
                if vi != vj and vj not in adj[vi]:
                    adj[vi].append(int(vj))
    return adj


def _py_taubin(v, adj, w, steps, dtype):
    v = v.astype(dtype).copy()
    for _ in range(steps):
        for k in (0.5, -0.53):
            d = np.zeros((len(v), 3), np.float64)
            for i, nb in enumerate(adj):
                acc = np.zeros(3, np.float64)
                for j in nb:
                    acc += v[i].astype(np.float64) - v[j].astype(np.float64)
                d[i] = acc / len(nb) if nb else acc
            for i in range(len(v)):
                v[i] += (w[i] * k * d[i]).astype(dtype)
    return v


def test_connectivity_order_matches_transcription(oracle):
    rng = np.random.default_rng(0)
    nv = 40
    f3 = rng.integers(0, nv, (120, 3))
    off, idx = oracle.mesh_vertex_connectivity(f3, nv)
    adj = _py_connectivity(f3, nv)
    for v in range(nv):
        assert list(idx[off[v]:off[v + 1]]) == adj[v]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_smoothing_matches_transcription_and_quirks(oracle, dtype):
    v, f3 = _octa()
    v = (v * np.float32(1.37) + np.float32(0.11)).astype(dtype)
    f4 = np.concatenate([np.full((len(f3), 1), 3), f3], axis=1)
    nrm = oracle.mesh_face_normals(v, f3)
    got = v.copy()
    flags, w = oracle.context_aware_smoothing(got, f4, nrm, 0.7, 3.0, 0.5, 4, details=True)
    # Q-M2: every vertex with a face is a "staircase" vertex, so its weight is 1; the lone vertex keeps bmin
    assert list(flags) == [1, 1, 1, 1, 1, 1, 0]
    assert list(w) == [1.0] * 6 + [0.5]
    want = _py_taubin(v, _py_connectivity(f3, len(v)), w, 4, dtype)
    assert np.array_equal(got, want)
    assert np.array_equal(got[6], v[6]) and not np.array_equal(got[:6], v[:6])
    # d = mean(p_i - p_j) points AWAY from the neighbours, so lambda = +0.5 inflates and mu = -0.53 shrinks
    # (mesh.rs:360-392): the symmetric octahedron stays centred and contracts
    assert np.allclose(got[:6].mean(0), v[:6].mean(0), atol=1e-5)
    r0, r1 = np.linalg.norm(v[:6] - v[:6].mean(0), axis=1).mean(), np.linalg.norm(got[:6] - got[:6].mean(0), axis=1).mean()
    assert r1 < r0


def test_count_column_files_all_faces_under_vertex_3(oracle):
    """Q-M1: vertex id 3 is flagged even when no triangle uses it"""
    v = np.zeros((6, 3), np.float32)
    v[:, 0] = np.arange(6)
    f4 = np.array([[3, 0, 1, 2], [3, 2, 4, 5]], np.int64)
    nrm = np.array([[0, 0, 1.0], [0, 0, 1.0]])
    flags, w = oracle.context_aware_smoothing(v.copy(), f4, nrm, 0.7, 3.0, 0.5, 0, details=True)
    assert list(flags) == [1, 1, 1, 1, 1, 1]
    assert w[3] == 1.0


def test_propagate_weights_partial_seeds(oracle):
    # a strip of vertices 1 apart: weights fall off linearly from the seed and stop at tmax
    n = 12
    v = np.zeros((n, 3), np.float64)
    v[:, 0] = np.arange(n)
    f3 = np.array([[i, i + 1, i + 1] for i in range(n - 1)])  # degenerate triangles still connect i -- i+1
    seeds = np.zeros(n, np.uint8)
    seeds[2] = 1
    w = oracle.mesh_propagate_weights(v, f3, seeds, 3.5, 0.25)
    d = np.abs(np.arange(n) - 2.0)
    want = np.where(d <= 3.5, (1 - d / 3.5) * 0.75 + 0.25, 0.25)
    assert np.allclose(w, want, rtol=0, atol=1e-15)
