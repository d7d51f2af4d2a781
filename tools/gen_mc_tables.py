#!/usr/bin/env python3
"""Generate the marching-cubes case tables used by BOTH the CPU oracle and the
HIP kernels (include/ivx_mc_tables.h).

Why generated, not transcribed: the reference extracts surfaces with VTK's
vtkContourFilter (invesalius/data/surface_process.py:172-184); VTK is not
vendored under the reference tree and is not installable here, so its private
case table cannot be consulted ("parity unpinned" for triangle topology, see
DESIGN.md).  What IS algorithm independent is the vertex set: every vertex is
a linear interpolation on a grid edge whose two end points straddle the
iso-value.  This script builds a watertight table from first principles:

  * corners  c = dx + 2*dy + 4*dz            (dx,dy,dz in {0,1})
  * edges    0..3  x-edges, (dy,dz) = (0,0),(1,0),(0,1),(1,1)
             4..7  y-edges, (dx,dz) = (0,0),(1,0),(0,1),(1,1)
             8..11 z-edges, (dx,dy) = (0,0),(1,0),(0,1),(1,1)
    every edge runs low corner -> high corner (canonical direction, so the
    interpolated vertex is bit-identical from all four cells sharing it).
  * a corner is "inside" when scalar >= iso (bit set in the case index).
  * on every cube face the active edges are joined by segments; an ambiguous
    face (two diagonal inside corners) always separates the INSIDE corners.
    The rule depends only on the four corner states of the face, so both
    cells sharing a face draw the same segments -> no cracks, ever.
  * segments chain into closed loops; each loop is fan-triangulated from its
    lowest-numbered edge and oriented so the normal points from inside
    (>= iso) to outside (< iso).

The emitted header holds: MC_EDGE_CORNERS[12][2], MC_EDGE_AXIS[12],
MC_EDGE_BASE[12][3] (low corner offset), MC_NTRI[256], MC_TRI[256][MAXT*3].
"""
import itertools
import sys

CORNER = [(c & 1, (c >> 1) & 1, (c >> 2) & 1) for c in range(8)]


def corner_id(p):
    return p[0] + 2 * p[1] + 4 * p[2]


EDGES = []  # (c_lo, c_hi, axis, base)
for axis in range(3):
    others = [a for a in range(3) if a != axis]
    for b in range(2):
        for a in range(2):
            base = [0, 0, 0]
            base[others[0]] = a
            base[others[1]] = b
            lo = tuple(base)
            hi = list(base)
            hi[axis] = 1
            EDGES.append((corner_id(lo), corner_id(tuple(hi)), axis, lo))
EDGE_OF = {}
for e, (a, b, _, _) in enumerate(EDGES):
    EDGE_OF[(a, b)] = e
    EDGE_OF[(b, a)] = e

# faces: list of 4 corners in cyclic order
FACES = []
for axis in range(3):
    o1, o2 = [a for a in range(3) if a != axis]
    for side in range(2):
        cyc = []
        for (u, v) in ((0, 0), (1, 0), (1, 1), (0, 1)):
            p = [0, 0, 0]
            p[axis] = side
            p[o1] = u
            p[o2] = v
            cyc.append(corner_id(tuple(p)))
        FACES.append(cyc)


def face_segments(case, cyc):
    ins = [(case >> c) & 1 for c in cyc]
    n = sum(ins)
    fe = [EDGE_OF[(cyc[i], cyc[(i + 1) % 4])] for i in range(4)]  # edge i joins corner i,i+1
    act = [ins[i] != ins[(i + 1) % 4] for i in range(4)]
    if n == 0 or n == 4:
        return []
    if n == 1 or n == 3:
        es = [fe[i] for i in range(4) if act[i]]
        assert len(es) == 2
        return [tuple(es)]
    # n == 2
    if ins[0] == ins[2]:  # diagonal -> ambiguous: separate the inside corners
        segs = []
        for i in range(4):
            if ins[i]:
                segs.append((fe[(i - 1) % 4], fe[i]))  # the two edges touching corner i
        return segs
    es = [fe[i] for i in range(4) if act[i]]
    assert len(es) == 2
    return [tuple(es)]


def edge_mid(e):
    a, b, _, _ = EDGES[e]
    pa, pb = CORNER[a], CORNER[b]
    return tuple((pa[i] + pb[i]) / 2.0 for i in range(3))


def loops_for_case(case):
    adj = {}
    for cyc in FACES:
        for (a, b) in face_segments(case, cyc):
            adj.setdefault(a, []).append(b)
            adj.setdefault(b, []).append(a)
    for e, nb in adj.items():
        assert len(nb) == 2, (case, e, nb)
    seen = set()
    loops = []
    for start in sorted(adj):
        if start in seen:
            continue
        loop = [start]
        seen.add(start)
        prev, cur = None, start
        while True:
            nxts = adj[cur]
            nxt = nxts[0] if nxts[0] != prev else nxts[1]
            if len(loop) > 1 and nxt == start:
                break
            if nxt in seen:
                # 2-cycle guard (cannot happen for a cube, kept for safety)
                break
            loop.append(nxt)
            seen.add(nxt)
            prev, cur = cur, nxt
        loops.append(loop)
    return loops


def orient(case, loop):
    pts = [edge_mid(e) for e in loop]
    n = [0.0, 0.0, 0.0]
    for i in range(len(pts)):
        p, q = pts[i], pts[(i + 1) % len(pts)]
        n[0] += (p[1] - q[1]) * (p[2] + q[2])
        n[1] += (p[2] - q[2]) * (p[0] + q[0])
        n[2] += (p[0] - q[0]) * (p[1] + q[1])
    d = [0.0, 0.0, 0.0]
    for e in loop:
        a, b, _, _ = EDGES[e]
        ia = (case >> a) & 1
        src, dst = (a, b) if ia else (b, a)  # inside -> outside
        for i in range(3):
            d[i] += CORNER[dst][i] - CORNER[src][i]
    dot = sum(n[i] * d[i] for i in range(3))
    assert abs(dot) > 1e-9, (case, loop)
    if dot < 0:
        loop = [loop[0]] + loop[1:][::-1]
    return loop


def build():
    tris = []
    for case in range(256):
        t = []
        for loop in loops_for_case(case):
            loop = orient(case, loop)
            for i in range(1, len(loop) - 1):
                t.append((loop[0], loop[i], loop[i + 1]))
        tris.append(t)
    return tris


def main(out_path):
    tris = build()
    maxt = max(len(t) for t in tris)
    lines = []
    w = lines.append
    w("/* GENERATED by tools/gen_mc_tables.py -- do not edit.")
    w(" * Marching-cubes case table shared by oracle/ivx_oracle.c and the HIP kernels.")
    w(" * Conventions: corner c = dx + 2*dy + 4*dz; inside <=> scalar >= iso;")
    w(" * edges 0-3 along x, 4-7 along y, 8-11 along z, each low->high corner. */")
    w("#ifndef IVX_MC_TABLES_H")
    w("#define IVX_MC_TABLES_H")
    w("#define MC_MAX_TRI %d" % maxt)
    w("#ifndef MC_TABLE_QUAL")
    w("#define MC_TABLE_QUAL static const")
    w("#endif")
    w("/* edge -> (low corner, high corner) */")
    w("MC_TABLE_QUAL unsigned char MC_EDGE_CORNERS[12][2] = {%s};"
      % ", ".join("{%d,%d}" % (a, b) for a, b, _, _ in EDGES))
    w("/* edge -> axis it runs along (0=x,1=y,2=z) */")
    w("MC_TABLE_QUAL unsigned char MC_EDGE_AXIS[12] = {%s};" % ", ".join(str(ax) for _, _, ax, _ in EDGES))
    w("/* edge -> (dx,dy,dz) of its low corner */")
    w("MC_TABLE_QUAL unsigned char MC_EDGE_BASE[12][3] = {%s};"
      % ", ".join("{%d,%d,%d}" % lo for _, _, _, lo in EDGES))
    w("MC_TABLE_QUAL unsigned char MC_NTRI[256] = {")
    for r in range(0, 256, 32):
        w("  " + ", ".join(str(len(tris[c])) for c in range(r, r + 32)) + ",")
    w("};")
    w("/* per case: MC_NTRI[case] triangles, 3 edge ids each, padded with 255 */")
    w("MC_TABLE_QUAL unsigned char MC_TRI[256][%d] = {" % (maxt * 3))
    for c in range(256):
        flat = [e for t in tris[c] for e in t]
        flat += [255] * (maxt * 3 - len(flat))
        w("  {" + ",".join("%d" % v for v in flat) + "},")
    w("};")
    w("#endif")
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote %s  (max triangles/cell = %d, total = %d)" % (out_path, maxt, sum(len(t) for t in tris)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "include/ivx_mc_tables.h")
