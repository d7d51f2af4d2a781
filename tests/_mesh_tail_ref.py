"""CHECKER (test infrastructure, never imported by the product): the rules of csrc/k_meshtail.hip -- the hole filling and the
point normals of join_process_surface (invesalius/data/surface_process.py:396-435: vtkFillHolesFilter with hole size 300,
vtkPolyDataNormals with feature angle 80, splitting, auto-orientation) -- stated in plain numpy / Python, convention for
convention, so that the kernels can be compared array for array:

* directed edge e = 3 f + k runs faces[f][k] -> faces[f][(k + 1) % 3]; of equal directed edges the smallest id counts;
* a rim (boundary) edge has no opposite edge; the successor of rim edge (a -> v) of face F is found by turning about v: F's
  next edge (v -> x), the face across it, that face's next edge, ... until an edge leaving v has no face across it (at most 64
  faces, else the chain ends) -- at a pinch point every fan of faces continues its own rim;
* a rim edge that is the successor of two rim edges (duplicated directed edges of non-manifold input: a chain hanging into a
  cycle) spoils every rim that reaches it -- such rims stay open, like chains that do not close;
* rims that close are ordered by their smallest edge and walked from it; a rim of >= 3 edges whose bounding sphere (half the
  diagonal of its bounding box) has a radius <= hole_size gets ONE new point, the mean of its points (float64, summed in
  walking order), and the triangles (b, a, centre) of its edges a -> b;
* normals: see `point_normals`.
Both VTK filters are third party and not installed: PARITY UNPINNED vs VTK, these rules are the documented behaviour."""
import numpy as np


def directed_edges(faces):
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    return np.stack([f, np.roll(f, -1, axis=1)], axis=2).reshape(-1, 2)  # row 3 f + k = (f[k], f[(k + 1) % 3])


def boundary_edges(faces):
    """indices (3 f + k) of the directed edges without an opposite edge"""
    e = directed_edges(faces)
    have = set(map(tuple, e.tolist()))
    return np.array([i for i, (a, b) in enumerate(e.tolist()) if (b, a) not in have], np.int64)


def rim_loops(faces):
    """-> list of closed rims, each a list of edge ids in walking order from the rim's smallest edge, ordered by that edge"""
    e = directed_edges(faces)
    rim = boundary_edges(faces)
    is_rim = set(rim.tolist())
    first = {}
    for i, (a, b) in enumerate(e.tolist()):
        first.setdefault((a, b), i)
    nxt = {}
    for i in rim.tolist():
        cur = i
        for _ in range(64):  # turn about the end point through the faces joined across its edges
            en = 3 * (cur // 3) + (cur % 3 + 1) % 3
            if en in is_rim:
                nxt[i] = en
                break
            v, x = e[en]
            cur = first.get((int(x), int(v)))
            if cur is None:
                break
    indeg = {}
    for j in nxt.values():
        indeg[j] = indeg.get(j, 0) + 1
    loops, seen = [], set()
    for i in rim.tolist():  # ascending: the first unseen edge of a closed rim is its smallest
        if i in seen:
            continue
        path, j = [], i
        while j is not None and j not in seen:
            seen.add(j)
            path.append(j)
            j = nxt.get(j)
        if j == i and all(indeg.get(k, 0) == 1 for k in path):  # a rim another chain hangs into is left open
            loops.append(path)
    return loops


def fill_holes(verts, faces, hole_size=300.0):
    """-> (verts, faces, holes filled); new points and triangles are appended"""
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    f = np.asarray(faces, np.int32).reshape(-1, 3)
    e = directed_edges(f)
    new_v, new_f = [], []
    for loop in rim_loops(f):
        if len(loop) < 3:
            continue
        pts = v[e[loop, 0]].astype(np.float64)
        d = pts.max(0) - pts.min(0)
        if 0.5 * np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > hole_size:
            continue
        s = np.zeros(3)
        for p in pts:  # (walking order, one addition at a time: the kernel's order)
            s = s + p
        c = len(v) + len(new_v)
        new_v.append((s / float(len(pts))).astype(np.float32))
        for i in loop:
            new_f.append((e[i, 1], e[i, 0], c))
    if not new_v:
        return v, f, 0
    return (np.concatenate([v, np.stack(new_v)]).astype(np.float32), np.concatenate([f, np.asarray(new_f, np.int32)]), len(new_v))


def point_normals(verts, faces, feature_angle=80.0, splitting=True, auto_orient=True):
    """vtkPolyDataNormals' documented behaviour: unit cell normals; points on an edge sharper than the feature angle are
    duplicated, one copy per fan of triangles joined by smooth edges (corners are joined across an edge when the two cell
    normals' dot product exceeds cos(feature angle)); the fan holding the vertex's smallest corner keeps the point, the
    copies are appended in (vertex, fan) order; a point's normal is the normalised sum of its fan's unit cell normals (corner
    order, float64); with auto-orientation a surface whose signed volume is negative is turned inside out first.
    -> (verts, faces, point normals float32, cell normals float32)"""
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    if not len(f):
        return v, f.astype(np.int32), np.zeros((len(v), 3), np.float32), np.zeros((0, 3), np.float32)
    p = v.astype(np.float64)
    if auto_orient:
        a, b, c = p[f[:, 0]], p[f[:, 1]], p[f[:, 2]]
        if float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) < 0.0:
            f = f[:, ::-1].copy()
    cn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    ln = np.sqrt((cn[:, 0] * cn[:, 0] + cn[:, 1] * cn[:, 1]) + cn[:, 2] * cn[:, 2])[:, None]
    cn = np.divide(cn, ln, out=np.zeros_like(cn), where=ln > 0)
    nf, nv = len(f), len(v)
    corner_v = f.reshape(-1)
    parent = np.arange(3 * nf)

    def find(x):
        while parent[x] != x:
            x = parent[x]
        return x

    if splitting:
        e = directed_edges(f)
        first = {}
        for i, (a, b) in enumerate(e.tolist()):
            first.setdefault((a, b), i)
        cosang = np.cos(np.deg2rad(feature_angle))
        for i, (a, b) in enumerate(e.tolist()):
            m = first.get((b, a))
            if m is None:
                continue
            fi, g = i // 3, m // 3
            d = (cn[fi, 0] * cn[g, 0] + cn[fi, 1] * cn[g, 1]) + cn[fi, 2] * cn[g, 2]
            if not d > cosang:
                continue
            for x, y in ((i, 3 * g + (m % 3 + 1) % 3), (3 * fi + (i % 3 + 1) % 3, m)):
                rx, ry = find(x), find(y)
                if rx != ry:
                    parent[max(rx, ry)] = min(rx, ry)
        label = np.array([find(c) for c in range(3 * nf)])
    else:
        firstc = np.full(nv, 3 * nf, np.int64)
        np.minimum.at(firstc, corner_v, np.arange(3 * nf))
        label = firstc[corner_v]
    out_f = np.empty(3 * nf, np.int64)
    extra_v, fan_corners = [], {}
    by_vertex = {}
    for c in range(3 * nf):
        by_vertex.setdefault(int(corner_v[c]), []).append((int(label[c]), c))
    nextra = 0
    for vtx in sorted(by_vertex):
        fans = {}
        for lab, c in sorted(by_vertex[vtx]):
            fans.setdefault(lab, []).append(c)
        for k, lab in enumerate(sorted(fans)):
            pid = vtx if k == 0 else nv + nextra
            if k:
                nextra += 1
                extra_v.append(v[vtx])
            fan_corners[pid] = fans[lab]
            out_f[fans[lab]] = pid
    out_v = np.concatenate([v, np.stack(extra_v)]) if extra_v else v
    pn = np.zeros((len(out_v), 3), np.float64)
    for pid, cs in fan_corners.items():
        s = np.zeros(3)
        for c in cs:
            s = s + cn[c // 3]
        l2 = np.sqrt((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2])
        pn[pid] = s / l2 if l2 > 0 else 0.0
    return out_v.astype(np.float32), out_f.reshape(-1, 3).astype(np.int32), pn.astype(np.float32), cn.astype(np.float32)
