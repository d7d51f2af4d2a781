/* ivx_oracle_mesh.c -- TEST INFRASTRUCTURE ONLY (see ivx_oracle.c header).
 *
 * CPU restatement of context-aware smoothing, invesalius_rs/src/mesh.rs:27-395 (entry: mesh_py.rs
 * context_aware_smoothing; caller: invesalius/data/surface_process.py:313-317).  The reference is Rust and there is
 * no Rust toolchain here, and its own test-suite holds no vectors for this function: PARITY UNPINNED beyond this
 * restatement, which follows the source statement by statement, quirks included:
 *   Q-M1  faces are (M,4) rows [3, v0, v1, v2] (vtkCellArray layout).  build_map_vface (mesh.rs:88-100) iterates the
 *         WHOLE row, so the leading count files every face under vertex id 3 as well.
 *   Q-M2  find_staircase_artifacts (mesh.rs:123-191) starts with max = f64::MIN, min = f64::MAX and updates min only
 *         in the `else` of the max update, so after the first incident face |max - min| ~ 1.8e308 >= t: every vertex
 *         that has at least one face is reported.  The loop is restated literally; nothing is special-cased.
 *   Q-M3  is_border() is the constant false (mesh.rs:331-338): Taubin smoothing treats every vertex as interior.
 * propagate_weights (mesh.rs:204-288) is a racy parallel relaxation in the reference (CAS on the distance, a separate
 * store of the seed); the schedule restated here is the synchronous one: every vertex of the frontier proposes with
 * the seed it held when the round began, each target keeps the smallest (distance, seed id) proposal.  With Q-M2 every
 * face vertex is its own seed at distance 0 and no proposal ever wins, so the schedule does not matter for the
 * results the reference can produce.
 * Vertices are float32 or float64 (V), arithmetic is float64 with one cast back to V per component and step, exactly
 * as the reference writes it (mesh.rs:360-392). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef struct {
    int64_t *off; /* nv + 1 */
    int64_t *idx;
} csr_t;

/* mesh.rs:88-100 -- entries per vertex in (face, position) order; position 0 is the count column */
static int build_map_vface(const int64_t *faces4, int64_t nt, int64_t nv, csr_t *m) {
    m->off = calloc((size_t)nv + 2, sizeof(int64_t));
    if (!m->off) return -1;
    for (int64_t f = 0; f < nt; f++)
        for (int q = 0; q < 4; q++) {
            const int64_t v = faces4[4 * f + q];
            if (v >= 0 && v < nv) m->off[v + 1]++;
        }
    for (int64_t v = 0; v < nv; v++) m->off[v + 1] += m->off[v];
    m->idx = malloc(sizeof(int64_t) * (size_t)(m->off[nv] + 1));
    int64_t *fill = malloc(sizeof(int64_t) * (size_t)(nv + 1));
    if (!m->idx || !fill) return -1;
    memcpy(fill, m->off, sizeof(int64_t) * (size_t)nv);
    for (int64_t f = 0; f < nt; f++)
        for (int q = 0; q < 4; q++) {
            const int64_t v = faces4[4 * f + q];
            if (v >= 0 && v < nv) m->idx[fill[v]++] = f;
        }
    free(fill);
    return 0;
}

/* mesh.rs:102-121 -- unique neighbours in order of first appearance */
static int build_vertex_connectivity(const int64_t *faces4, int64_t nt, int64_t nv, csr_t *adj) {
    /* capacity: two new neighbours per (face, corner) at most */
    int64_t *cap = calloc((size_t)nv + 2, sizeof(int64_t));
    if (!cap) return -1;
    for (int64_t f = 0; f < nt; f++)
        for (int q = 1; q < 4; q++) cap[faces4[4 * f + q] + 1] += 2;
    for (int64_t v = 0; v < nv; v++) cap[v + 1] += cap[v];
    int64_t *idx = malloc(sizeof(int64_t) * (size_t)(cap[nv] + 1));
    int64_t *deg = calloc((size_t)nv + 1, sizeof(int64_t));
    if (!idx || !deg) return -1;
    for (int64_t f = 0; f < nt; f++)
        for (int qi = 1; qi < 4; qi++)
            for (int qj = 1; qj < 4; qj++) {
                const int64_t vi = faces4[4 * f + qi], vj = faces4[4 * f + qj];
                if (vi == vj) continue;
                int found = 0;
                for (int64_t e = 0; e < deg[vi]; e++)
                    if (idx[cap[vi] + e] == vj) {
                        found = 1;
                        break;
                    }
                if (!found) idx[cap[vi] + deg[vi]++] = vj;
            }
    /* compact */
    adj->off = calloc((size_t)nv + 2, sizeof(int64_t));
    if (!adj->off) return -1;
    for (int64_t v = 0; v < nv; v++) adj->off[v + 1] = adj->off[v] + deg[v];
    adj->idx = malloc(sizeof(int64_t) * (size_t)(adj->off[nv] + 1));
    if (!adj->idx) return -1;
    for (int64_t v = 0; v < nv; v++) memcpy(adj->idx + adj->off[v], idx + cap[v], sizeof(int64_t) * (size_t)deg[v]);
    free(cap);
    free(idx);
    free(deg);
    return 0;
}

/* mesh.rs:123-191, literal */
static int is_staircase(const csr_t *m, int64_t v, const double *normals, const double so[3], double t) {
    double max_z = -DBL_MAX, min_z = DBL_MAX, max_y = -DBL_MAX, min_y = DBL_MAX, max_x = -DBL_MAX, min_x = DBL_MAX;
    for (int64_t e = m->off[v]; e < m->off[v + 1]; e++) {
        const double *n = normals + 3 * m->idx[e];
        const double of_z = 1.0 - fabs(n[0] * so[0] + n[1] * so[1] + n[2] * so[2]);
        const double of_y = 1.0 - fabs(n[0] * 0.0 + n[1] * 1.0 + n[2] * 0.0);
        const double of_x = 1.0 - fabs(n[0] * 1.0 + n[1] * 0.0 + n[2] * 0.0);
        if (of_z > max_z) max_z = of_z;
        else if (of_z < min_z) min_z = of_z;
        if (of_y > max_y) max_y = of_y;
        else if (of_y < min_y) min_y = of_y;
        if (of_x > max_x) max_x = of_x;
        else if (of_x < min_x) min_x = of_x;
        if (fabs(max_z - min_z) >= t || fabs(max_y - min_y) >= t || fabs(max_x - min_x) >= t) return 1;
    }
    return 0;
}

static inline double vget(const void *verts, int is64, int64_t v, int c) {
    return is64 ? ((const double *)verts)[3 * v + c] : (double)((const float *)verts)[3 * v + c];
}

/* mesh.rs:204-288, synchronous schedule (see header).  seeds: flag per vertex.  out: weights[nv] */
int orc_mesh_propagate_weights(const void *verts, int is64, int64_t nv, const int64_t *adj_off, const int64_t *adj_idx,
                               const uint8_t *seed_flag, double tmax, double bmin, double *weights) {
    double *dist = malloc(sizeof(double) * (size_t)(nv + 1)), *nd = malloc(sizeof(double) * (size_t)(nv + 1));
    int64_t *seed = malloc(sizeof(int64_t) * (size_t)(nv + 1)), *ns = malloc(sizeof(int64_t) * (size_t)(nv + 1));
    uint8_t *front = calloc((size_t)nv + 1, 1), *nf = calloc((size_t)nv + 1, 1);
    if (!dist || !nd || !seed || !ns || !front || !nf) return -1;
    int64_t nfront = 0;
    for (int64_t v = 0; v < nv; v++) {
        dist[v] = seed_flag[v] ? 0.0 : INFINITY;
        seed[v] = seed_flag[v] ? v : -1;
        front[v] = seed_flag[v];
        nfront += seed_flag[v];
    }
    const double tmax_sq = tmax * tmax;
    while (nfront) {
        memcpy(nd, dist, sizeof(double) * (size_t)nv);
        memcpy(ns, seed, sizeof(int64_t) * (size_t)nv);
        memset(nf, 0, (size_t)nv);
        for (int64_t v = 0; v < nv; v++) {
            if (!front[v]) continue;
            const int64_t s = seed[v];
            for (int64_t e = adj_off[v]; e < adj_off[v + 1]; e++) {
                const int64_t vj = adj_idx[e];
                const double dx = vget(verts, is64, vj, 0) - vget(verts, is64, s, 0);
                const double dy = vget(verts, is64, vj, 1) - vget(verts, is64, s, 1);
                const double dz = vget(verts, is64, vj, 2) - vget(verts, is64, s, 2);
                const double d_sq = dx * dx + dy * dy + dz * dz;
                if (d_sq > tmax_sq) continue;
                if (!(d_sq < dist[vj])) continue; /* d_sq >= old: no update (old finite or not: inf never loses) */
                if (d_sq < nd[vj] || (d_sq == nd[vj] && nf[vj] && s < ns[vj])) {
                    nd[vj] = d_sq;
                    ns[vj] = s;
                    nf[vj] = 1;
                }
            }
        }
        nfront = 0;
        for (int64_t v = 0; v < nv; v++) {
            dist[v] = nd[v];
            seed[v] = ns[v];
            front[v] = nf[v];
            nfront += nf[v];
        }
    }
    for (int64_t v = 0; v < nv; v++) {
        const double d = dist[v];
        weights[v] = isfinite(d) ? (1.0 - sqrt(d) / tmax) * (1.0 - bmin) + bmin : bmin;
    }
    free(dist); free(nd); free(seed); free(ns); free(front); free(nf);
    return 0;
}

/* mesh.rs:290-329 with is_border == false */
static void calc_d(const void *verts, int is64, const csr_t *adj, int64_t v, double d[3]) {
    const double px = vget(verts, is64, v, 0), py = vget(verts, is64, v, 1), pz = vget(verts, is64, v, 2);
    d[0] = d[1] = d[2] = 0.0;
    int64_t n = 0;
    for (int64_t e = adj->off[v]; e < adj->off[v + 1]; e++) {
        const int64_t vj = adj->idx[e];
        d[0] += px - vget(verts, is64, vj, 0);
        d[1] += py - vget(verts, is64, vj, 1);
        d[2] += pz - vget(verts, is64, vj, 2);
        n++;
    }
    if (n > 0) {
        d[0] /= (double)n;
        d[1] /= (double)n;
        d[2] /= (double)n;
    }
}

/* mesh.rs:340-395 */
static int taubin_smooth(void *verts, int is64, int64_t nv, const csr_t *adj, const double *w, double l, double m,
                         int steps) {
    double *dv = malloc(sizeof(double) * 3 * (size_t)(nv + 1));
    if (!dv) return -1;
    for (int s = 0; s < steps; s++)
        for (int half = 0; half < 2; half++) {
            const double k = half ? m : l;
            for (int64_t v = 0; v < nv; v++) calc_d(verts, is64, adj, v, dv + 3 * v);
            for (int64_t v = 0; v < nv; v++)
                for (int c = 0; c < 3; c++) {
                    const double step = w[v] * k * dv[3 * v + c];
                    /* NumCast::from(f64) -> V never fails for floats (inf/NaN pass through) */
                    if (is64) ((double *)verts)[3 * v + c] += step;
                    else ((float *)verts)[3 * v + c] += (float)step;
                }
        }
    free(dv);
    return 0;
}

/* context_aware_smoothing_internal, mesh.rs:27-86.  faces4: (nt,4) int64 rows [3,v0,v1,v2]; normals: (nt,3) f64.
 * Optional outputs (may be NULL): staircase[nv] flags, weights[nv]. */
int orc_context_aware_smoothing(void *verts, int is64, int64_t nv, const int64_t *faces4, int64_t nt,
                                const double *normals, double t, double tmax, double bmin, int n_iters,
                                uint8_t *staircase_out, double *weights_out) {
    for (int64_t f = 0; f < nt; f++)
        for (int q = 1; q < 4; q++)
            if (faces4[4 * f + q] < 0 || faces4[4 * f + q] >= nv) return -2;
    csr_t map = {0, 0}, adj = {0, 0};
    if (build_map_vface(faces4, nt, nv, &map)) return -1;
    if (build_vertex_connectivity(faces4, nt, nv, &adj)) return -1;
    const double so[3] = {0.0, 0.0, 1.0};
    uint8_t *flag = calloc((size_t)nv + 1, 1);
    double *w = malloc(sizeof(double) * (size_t)(nv + 1));
    if (!flag || !w) return -1;
    for (int64_t v = 0; v < nv; v++) flag[v] = (uint8_t)is_staircase(&map, v, normals, so, t);
    if (orc_mesh_propagate_weights(verts, is64, nv, adj.off, adj.idx, flag, tmax, bmin, w)) return -1;
    if (staircase_out) memcpy(staircase_out, flag, (size_t)nv);
    if (weights_out) memcpy(weights_out, w, sizeof(double) * (size_t)nv);
    if (taubin_smooth(verts, is64, nv, &adj, w, 0.5, -0.53, n_iters)) return -1;
    free(map.off); free(map.idx); free(adj.off); free(adj.idx); free(flag); free(w);
    return 0;
}

/* adjacency in the reference's order, for tests: call with idx == NULL to get the sizes (off[nv] = total) */
int orc_mesh_vertex_connectivity(const int64_t *faces4, int64_t nt, int64_t nv, int64_t *off, int64_t *idx) {
    csr_t adj = {0, 0};
    if (build_vertex_connectivity(faces4, nt, nv, &adj)) return -1;
    memcpy(off, adj.off, sizeof(int64_t) * (size_t)(nv + 1));
    if (idx) memcpy(idx, adj.idx, sizeof(int64_t) * (size_t)adj.off[nv]);
    free(adj.off); free(adj.idx);
    return 0;
}
