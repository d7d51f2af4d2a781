// k_smooth.hip -- context-aware smoothing of the indexed surface.
//
// Reference (bit-exact target): invesalius_rs/src/mesh.rs:27-395 (context_aware_smoothing_internal), bound by
// mesh_py.rs and called from join_process_surface, invesalius/data/surface_process.py:313-317.  Stages, as there:
//   build_map_vface          mesh.rs:88-100    vertex -> incident faces, in (face, position) order
//   build_vertex_connectivity mesh.rs:102-121  vertex -> unique neighbours, in order of first appearance
//   find_staircase_artifacts mesh.rs:123-191   seed vertices (restated literally, quirks Q-M1/Q-M2 of
//                                              oracle/ivx_oracle_mesh.c included)
//   propagate_weights        mesh.rs:204-288   frontier relaxation of squared distance to the nearest seed
//   taubin_smooth            mesh.rs:340-395   2 * n_iters Jacobi half-steps, lambda 0.5 / mu -0.53
//
// MI355X design.  Everything is gather/scatter over a mesh that fits L2 + Infinity Cache (3.3 M vertices = 40 MB of
// float32 positions at 512^3); HBM-bound, no MFMA.
//   * The neighbour ORDER fixes the float64 summation order of every Laplacian, so adjacency is built
//     deterministically: incident faces per vertex by counting + scan + atomic fill, then each vertex sorts its own
//     short list and de-duplicates neighbours walking faces in ascending order -- the reference's global scan order,
//     recovered per vertex with no serial pass.
//   * Smoothing is ping-pong (read buffer A, write buffer B): one kernel per half-step instead of the reference's
//     D-array pass + update pass; 2 * n_iters is even, so the result lands back in the caller's buffer.
//   * propagate_weights runs the synchronous schedule of the reference's (racy) relaxation: distance bits as uint64
//     (non-negative doubles order like integers) through atomicMin, then the smallest seed id among the winners.
#include <algorithm>

#include "ivx_internal.h"
#include "scan_u32.h"

namespace {

template <typename V>
__device__ __forceinline__ double vx(const V *__restrict__ p, int64_t v, int c) { return (double)p[3 * v + c]; }

// ---- incident faces -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sm_zero(uint32_t *__restrict__ a, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = 0u;
}
// the counting pass keeps what atomicAdd returns (the entry's slot inside its vertex' list): the fill is then a plain scatter
__global__ __launch_bounds__(256) void k_sm_count(const int32_t *__restrict__ faces, int64_t nt, uint32_t *__restrict__ cnt,
                                                  uint32_t *__restrict__ slot) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * nt) slot[i] = atomicAdd(&cnt[(uint32_t)faces[i]], 1u);
}
__global__ __launch_bounds__(256) void k_sm_fill(const int32_t *__restrict__ faces, int64_t nt, const uint32_t *__restrict__ foff,
                                                 const uint32_t *__restrict__ slot, uint32_t *__restrict__ inc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nt) return;
    inc[foff[(uint32_t)faces[i]] + slot[i]] = (uint32_t)(i / 3);
}

// one lane per vertex: sort its incident faces, then list unique neighbours in the reference's order.
// Lists of up to RL entries (practically all of a marching-cubes surface: valence ~6) live in registers -- every loop
// below is fully unrolled with constant indices -- longer ones spill over to their global slots.
constexpr int RL = 12;
__global__ __launch_bounds__(256) void k_sm_adjacency(const int32_t *__restrict__ faces, int64_t nv,
                                                      const uint32_t *__restrict__ foff, uint32_t *__restrict__ inc,
                                                      uint32_t *__restrict__ adjpad, uint32_t *__restrict__ deg) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nv) return;
    if (v == nv) {
        deg[v] = 0u;
        return;
    }
    const uint32_t b = foff[v], e = foff[v + 1], len = e - b;
    uint32_t *out = adjpad + 2 * (int64_t)b;
    uint32_t n = 0;
    uint32_t nb[RL];
    auto add_face = [&](uint32_t f) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const uint32_t vj = (uint32_t)faces[3 * (int64_t)f + q];
            if (vj == (uint32_t)v) continue;
            bool found = false;
#pragma unroll
            for (int k = 0; k < RL; k++) found |= (uint32_t)k < n && nb[k] == vj;
            for (uint32_t k = RL; k < n; k++) found |= out[k] == vj;
            if (found) continue;
#pragma unroll
            for (int k = 0; k < RL; k++)
                if ((uint32_t)k == n) nb[k] = vj;
            if (n >= (uint32_t)RL) out[n] = vj;
            n++;
        }
    };
    if (len <= (uint32_t)RL) {
        uint32_t fl[RL];
#pragma unroll
        for (int k = 0; k < RL; k++) fl[k] = (uint32_t)k < len ? inc[b + k] : 0xffffffffu;
        // odd-even transposition sort: RL passes of constant-index compare-exchanges (padding sorts to the end)
#pragma unroll
        for (int pass = 0; pass < RL; pass++) {
#pragma unroll
            for (int k = pass & 1; k + 1 < RL; k += 2) {
                const uint32_t lo = fl[k] < fl[k + 1] ? fl[k] : fl[k + 1];
                const uint32_t hi = fl[k] < fl[k + 1] ? fl[k + 1] : fl[k];
                fl[k] = lo;
                fl[k + 1] = hi;
            }
        }
        uint32_t last = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < RL; k++) {
            if ((uint32_t)k < len) {
                inc[b + k] = fl[k]; // the staircase pass reads the sorted list
                if (fl[k] != last) add_face(fl[k]); // a face listed twice (degenerate triangle) is walked once
                last = fl[k];
            }
        }
    } else {
        for (uint32_t i = b + 1; i < e; i++) { // insertion sort in place
            const uint32_t x = inc[i];
            uint32_t j = i;
            while (j > b && inc[j - 1] > x) {
                inc[j] = inc[j - 1];
                j--;
            }
            inc[j] = x;
        }
        uint32_t last = 0xffffffffu;
        for (uint32_t i = b; i < e; i++) {
            const uint32_t f = inc[i];
            if (f != last) add_face(f);
            last = f;
        }
    }
#pragma unroll
    for (int k = 0; k < RL; k++)
        if ((uint32_t)k < n) out[k] = nb[k];
    deg[v] = n;
}
__global__ __launch_bounds__(256) void k_sm_compact_adj(int64_t nv, const uint32_t *__restrict__ foff,
                                                        const uint32_t *__restrict__ adjpad, const uint32_t *__restrict__ aoff,
                                                        uint32_t *__restrict__ adj) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const uint32_t a = aoff[v], n = aoff[v + 1] - a;
    const uint32_t *src = adjpad + 2 * (int64_t)foff[v];
    for (uint32_t k = 0; k < n; k++) adj[a + k] = src[k];
}

// ---- find_staircase_artifacts, literal ------------------------------------------------------------------------------
struct Stair {
    double max_z, min_z, max_y, min_y, max_x, min_x;
};
__device__ __forceinline__ bool stair_step(Stair &s, const double *__restrict__ n, double so0, double so1, double so2,
                                           double t) {
    const double of_z = 1.0 - fabs(n[0] * so0 + n[1] * so1 + n[2] * so2);
    const double of_y = 1.0 - fabs(n[0] * 0.0 + n[1] * 1.0 + n[2] * 0.0);
    const double of_x = 1.0 - fabs(n[0] * 1.0 + n[1] * 0.0 + n[2] * 0.0);
    if (of_z > s.max_z) s.max_z = of_z;
    else if (of_z < s.min_z) s.min_z = of_z;
    if (of_y > s.max_y) s.max_y = of_y;
    else if (of_y < s.min_y) s.min_y = of_y;
    if (of_x > s.max_x) s.max_x = of_x;
    else if (of_x < s.min_x) s.min_x = of_x;
    return fabs(s.max_z - s.min_z) >= t || fabs(s.max_y - s.min_y) >= t || fabs(s.max_x - s.min_x) >= t;
}
__global__ __launch_bounds__(256) void k_sm_staircase(int64_t nv, int64_t nt, const uint32_t *__restrict__ foff,
                                                      const uint32_t *__restrict__ inc, const double *__restrict__ normals,
                                                      double t, uint8_t *__restrict__ flag) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    constexpr double DMAX = 1.7976931348623157e308;
    Stair s = {-DMAX, DMAX, -DMAX, DMAX, -DMAX, DMAX};
    const uint32_t b = foff[v], e = foff[v + 1];
    bool hit = false;
    if (v != 3) {
        for (uint32_t i = b; i < e && !hit; i++) hit = stair_step(s, normals + 3 * (int64_t)inc[i], 0.0, 0.0, 1.0, t);
    } else {
        // Q-M1: the count column of the (M,4) face rows files EVERY face under vertex id 3, ahead of its real entries
        uint32_t i = b;
        for (int64_t f = 0; f < nt && !hit; f++) {
            hit = stair_step(s, normals + 3 * f, 0.0, 0.0, 1.0, t);
            while (!hit && i < e && inc[i] == (uint32_t)f) {
                hit = stair_step(s, normals + 3 * f, 0.0, 0.0, 1.0, t);
                i++;
            }
        }
    }
    flag[v] = hit ? 1 : 0;
}

// ---- propagate_weights, synchronous schedule ---------------------------------------------------------------------------
constexpr unsigned long long INF_BITS = 0x7ff0000000000000ull;
__global__ __launch_bounds__(256) void k_pw_init(int64_t nv, const uint8_t *__restrict__ flag,
                                                 unsigned long long *__restrict__ dist, uint32_t *__restrict__ seed,
                                                 uint8_t *__restrict__ front) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const bool s = flag[v] != 0;
    dist[v] = s ? 0ull : INF_BITS;
    seed[v] = s ? (uint32_t)v : 0xffffffffu;
    front[v] = s ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_pw_begin(int64_t nv, const unsigned long long *__restrict__ dist,
                                                  unsigned long long *__restrict__ nd, uint32_t *__restrict__ ns,
                                                  uint8_t *__restrict__ nf, uint32_t *__restrict__ changed) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) *changed = 0u;
    if (v >= nv) return;
    nd[v] = dist[v];
    ns[v] = 0xffffffffu;
    nf[v] = 0;
}
template <typename V, int PHASE>
__global__ __launch_bounds__(256) void k_pw_relax(const V *__restrict__ pos, int64_t nv, const uint32_t *__restrict__ aoff,
                                                  const uint32_t *__restrict__ adj, const uint8_t *__restrict__ front,
                                                  const unsigned long long *__restrict__ dist,
                                                  const uint32_t *__restrict__ seed, double tmax_sq,
                                                  unsigned long long *__restrict__ nd, uint32_t *__restrict__ ns,
                                                  uint8_t *__restrict__ nf) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv || !front[v]) return;
    const uint32_t s = seed[v];
    const double sx = vx(pos, s, 0), sy = vx(pos, s, 1), sz = vx(pos, s, 2);
    for (uint32_t e = aoff[v]; e < aoff[v + 1]; e++) {
        const uint32_t vj = adj[e];
        const double dx = vx(pos, vj, 0) - sx, dy = vx(pos, vj, 1) - sy, dz = vx(pos, vj, 2) - sz;
        const double d_sq = dx * dx + dy * dy + dz * dz;
        if (d_sq > tmax_sq) continue;
        if (!(d_sq < __longlong_as_double((long long)dist[vj]))) continue;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(d_sq);
        if (PHASE == 0) {
            atomicMin(&nd[vj], bits);
        } else if (nd[vj] == bits) {
            atomicMin(&ns[vj], s);
            nf[vj] = 1;
        }
    }
}
__global__ __launch_bounds__(256) void k_pw_commit(int64_t nv, unsigned long long *__restrict__ dist,
                                                   uint32_t *__restrict__ seed, uint8_t *__restrict__ front,
                                                   const unsigned long long *__restrict__ nd, const uint32_t *__restrict__ ns,
                                                   const uint8_t *__restrict__ nf, uint32_t *__restrict__ changed) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool c = v < nv && nf[v];
    if (v < nv) {
        front[v] = c ? 1 : 0;
        if (c) {
            dist[v] = nd[v];
            seed[v] = ns[v];
        }
    }
    const unsigned long long any = __ballot(c);
    if (any && (threadIdx.x & 63) == 0) atomicAdd(changed, (uint32_t)__popcll(any));
}
__global__ __launch_bounds__(256) void k_pw_weights(int64_t nv, const unsigned long long *__restrict__ dist, double tmax,
                                                    double bmin, double *__restrict__ w) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const double d = __longlong_as_double((long long)dist[v]);
    w[v] = isfinite(d) ? (1.0 - sqrt(d) / tmax) * (1.0 - bmin) + bmin : bmin;
}

// ---- one Taubin half-step: dst = src + V(w * k * mean_j(src_i - src_j)) ----------------------------------------------------
template <typename V>
__global__ __launch_bounds__(256) void k_taubin(const V *__restrict__ src, V *__restrict__ dst, int64_t nv,
                                                const uint32_t *__restrict__ aoff, const uint32_t *__restrict__ adj,
                                                const double *__restrict__ w, double k) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const V p0 = src[3 * v], p1 = src[3 * v + 1], p2 = src[3 * v + 2];
    const double px = (double)p0, py = (double)p1, pz = (double)p2;
    const uint32_t b = aoff[v], e = aoff[v + 1];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    for (uint32_t i = b; i < e; i++) {
        const int64_t vj = adj[i];
        d0 += px - (double)src[3 * vj];
        d1 += py - (double)src[3 * vj + 1];
        d2 += pz - (double)src[3 * vj + 2];
    }
    if (e > b) {
        const double n = (double)(e - b);
        d0 /= n;
        d1 /= n;
        d2 /= n;
    }
    const double wk = w[v] * k; // (weights[i] * l) * d, left to right as the reference writes it
    dst[3 * v] = p0 + (V)(wk * d0);
    dst[3 * v + 1] = p1 + (V)(wk * d1);
    dst[3 * v + 2] = p2 + (V)(wk * d2);
}

// face normals the way vtkPolyDataNormals' cell normals are formed (vtkPolygon::ComputeNormal: normalised
// (p1-p0)x(p2-p0) in double; zero for a degenerate triangle)
template <typename V>
__global__ __launch_bounds__(256) void k_face_normals(const V *__restrict__ pos, const int32_t *__restrict__ faces, int64_t nt,
                                                      double *__restrict__ normals) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nt) return;
    const int64_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
    const double ax = vx(pos, b, 0) - vx(pos, a, 0), ay = vx(pos, b, 1) - vx(pos, a, 1), az = vx(pos, b, 2) - vx(pos, a, 2);
    const double bx = vx(pos, c, 0) - vx(pos, a, 0), by = vx(pos, c, 1) - vx(pos, a, 1), bz = vx(pos, c, 2) - vx(pos, a, 2);
    double n0 = ay * bz - az * by, n1 = az * bx - ax * bz, n2 = ax * by - ay * bx;
    const double len = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    if (len != 0.0) {
        n0 /= len;
        n1 /= len;
        n2 /= len;
    }
    normals[3 * f] = n0;
    normals[3 * f + 1] = n1;
    normals[3 * f + 2] = n2;
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct SmLayout {
    size_t foff, cursor, inc, adjpad, aoff, adj, dist, nd, seed, ns, front, nf, flag, w, tmp, bsum, misc, total;
};
static SmLayout sm_layout(int64_t nv, int64_t nt, size_t vsize) {
    SmLayout m;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += al256(bytes + 16);
        return at;
    };
    m.foff = take(((size_t)nv + 2) * 4);
    m.cursor = take(((size_t)nv + 2) * 4);
    m.inc = take((size_t)nt * 3 * 4);
    m.adjpad = take((size_t)nt * 6 * 4);
    m.aoff = take(((size_t)nv + 2) * 4);
    m.adj = take((size_t)nt * 6 * 4);
    m.dist = take((size_t)nv * 8);
    m.nd = take((size_t)nv * 8);
    m.seed = take((size_t)nv * 4);
    m.ns = take((size_t)nv * 4);
    m.front = take((size_t)nv);
    m.nf = take((size_t)nv);
    m.flag = take((size_t)nv);
    m.w = take((size_t)nv * 8);
    m.tmp = take((size_t)nv * 3 * vsize);
    m.bsum = take(((size_t)scan_u32_blocks(nv + 2) + 16) * 4);
    m.misc = take(256);
    m.total = o;
    return m;
}

struct SmBuffers {
    uint32_t *foff, *cursor, *inc, *adjpad, *aoff, *adj, *seed, *ns, *bsum, *misc;
    unsigned long long *dist, *nd;
    uint8_t *front, *nf, *flag;
    double *w;
    void *tmp;
};
static int sm_buffers(int64_t nv, int64_t nt, size_t vsize, hipStream_t st, SmBuffers *b) {
    const SmLayout m = sm_layout(nv, nt, vsize);
    void *ws;
    int rc = ivx::ws_get_s(ivx::WS_MESH, st, m.total, &ws);
    if (rc) return rc;
    char *w = (char *)ws;
    b->foff = (uint32_t *)(w + m.foff); b->cursor = (uint32_t *)(w + m.cursor); b->inc = (uint32_t *)(w + m.inc);
    b->adjpad = (uint32_t *)(w + m.adjpad); b->aoff = (uint32_t *)(w + m.aoff); b->adj = (uint32_t *)(w + m.adj);
    b->dist = (unsigned long long *)(w + m.dist); b->nd = (unsigned long long *)(w + m.nd);
    b->seed = (uint32_t *)(w + m.seed); b->ns = (uint32_t *)(w + m.ns);
    b->front = (uint8_t *)(w + m.front); b->nf = (uint8_t *)(w + m.nf); b->flag = (uint8_t *)(w + m.flag);
    b->w = (double *)(w + m.w); b->tmp = w + m.tmp; b->bsum = (uint32_t *)(w + m.bsum); b->misc = (uint32_t *)(w + m.misc);
    return IVX_OK;
}

static inline unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, ivx::cdiv(n, 256)); }

// incident faces (sorted) + adjacency in the reference's order -> b.foff/inc, b.aoff/adj
static int build_topology(const int32_t *faces, int64_t nv, int64_t nt, const SmBuffers &b, hipStream_t st) {
    int rc;
    hipLaunchKernelGGL(k_sm_zero, dim3(std::min(grid_for(nv + 2), 16384u)), dim3(256), 0, st, b.foff, nv + 2);
    IVX_LAUNCH_CHECK();
    uint32_t *slot = b.adj; // free until k_sm_compact_adj writes the adjacency
    if (nt) {
        hipLaunchKernelGGL(k_sm_count, dim3(grid_for(3 * nt)), dim3(256), 0, st, faces, nt, b.foff, slot);
        IVX_LAUNCH_CHECK();
    }
    if ((rc = scan_u32_exclusive(b.foff, nv + 1, b.bsum, b.misc + 8, st))) return rc;
    if (nt) {
        hipLaunchKernelGGL(k_sm_fill, dim3(grid_for(3 * nt)), dim3(256), 0, st, faces, nt, b.foff, slot, b.inc);
        IVX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_sm_adjacency, dim3(grid_for(nv + 1)), dim3(256), 0, st, faces, nv, b.foff, b.inc, b.adjpad, b.aoff);
    IVX_LAUNCH_CHECK();
    if ((rc = scan_u32_exclusive(b.aoff, nv + 1, b.bsum, b.misc + 9, st))) return rc;
    hipLaunchKernelGGL(k_sm_compact_adj, dim3(grid_for(nv)), dim3(256), 0, st, nv, b.foff, b.adjpad, b.aoff, b.adj);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

template <typename V>
static int propagate(const V *verts, int64_t nv, const SmBuffers &b, double tmax, double bmin, hipStream_t st) {
    const unsigned g = grid_for(nv);
    hipLaunchKernelGGL(k_pw_init, dim3(g), dim3(256), 0, st, nv, b.flag, b.dist, b.seed, b.front);
    IVX_LAUNCH_CHECK();
    const double tmax_sq = tmax * tmax;
    uint32_t *changed = b.misc;
    for (int64_t round = 0; round <= nv; round++) { // a vertex's distance strictly decreases: it terminates
        hipLaunchKernelGGL(k_pw_begin, dim3(g), dim3(256), 0, st, nv, b.dist, b.nd, b.ns, b.nf, changed);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_pw_relax<V, 0>), dim3(g), dim3(256), 0, st, verts, nv, b.aoff, b.adj, b.front, b.dist, b.seed,
                           tmax_sq, b.nd, b.ns, b.nf);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_pw_relax<V, 1>), dim3(g), dim3(256), 0, st, verts, nv, b.aoff, b.adj, b.front, b.dist, b.seed,
                           tmax_sq, b.nd, b.ns, b.nf);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_pw_commit, dim3(g), dim3(256), 0, st, nv, b.dist, b.seed, b.front, b.nd, b.ns, b.nf, changed);
        IVX_LAUNCH_CHECK();
        uint32_t seq, got = 0;
        int rc;
        if ((rc = ivx::mailbox_publish(changed, 1, st, &seq))) return rc;
        if ((rc = ivx::mailbox_wait(seq, st, &got, 1))) return rc;
        if (!got) break;
    }
    hipLaunchKernelGGL(k_pw_weights, dim3(g), dim3(256), 0, st, nv, b.dist, tmax, bmin, b.w);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

template <typename V>
static int run_ca(V *verts, int64_t nv, const int32_t *faces, int64_t nt, const double *normals, double t, double tmax,
                  double bmin, int n_iters, const uint8_t *seeds_in, uint8_t *stair_out, double *w_out, bool smooth,
                  hipStream_t st) {
    SmBuffers b;
    int rc;
    if ((rc = sm_buffers(nv, nt, sizeof(V), st, &b))) return rc;
    if ((rc = build_topology(faces, nv, nt, b, st))) return rc;
    if (seeds_in) {
        IVX_HIP(hipMemcpyAsync(b.flag, seeds_in, (size_t)nv, hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(k_sm_staircase, dim3(grid_for(nv)), dim3(256), 0, st, nv, nt, b.foff, b.inc, normals, t, b.flag);
        IVX_LAUNCH_CHECK();
    }
    if ((rc = propagate<V>(verts, nv, b, tmax, bmin, st))) return rc;
    if (stair_out) IVX_HIP(hipMemcpyAsync(stair_out, b.flag, (size_t)nv, hipMemcpyDeviceToDevice, st));
    if (w_out) IVX_HIP(hipMemcpyAsync(w_out, b.w, (size_t)nv * 8, hipMemcpyDeviceToDevice, st));
    if (!smooth) return IVX_OK;
    V *tmp = (V *)b.tmp;
    for (int s = 0; s < n_iters; s++) {
        hipLaunchKernelGGL((k_taubin<V>), dim3(grid_for(nv)), dim3(256), 0, st, (const V *)verts, tmp, nv, b.aoff, b.adj, b.w, 0.5);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_taubin<V>), dim3(grid_for(nv)), dim3(256), 0, st, (const V *)tmp, verts, nv, b.aoff, b.adj, b.w,
                           -0.53);
        IVX_LAUNCH_CHECK();
    }
    return IVX_OK;
}

} // namespace

extern "C" int ivx_dev_context_aware_smoothing(void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                                               const double *normals, double t, double tmax, double bmin, int n_iters,
                                               uint8_t *staircase_out, double *weights_out, void *stream) {
    IVX_REQUIRE(vdtype == IVX_F32 || vdtype == IVX_F64, IVX_EINVAL, "ca_smoothing: vertices must be float32 or float64");
    IVX_REQUIRE(nverts >= 0 && ntris >= 0 && n_iters >= 0, IVX_EINVAL, "ca_smoothing: negative size");
    IVX_REQUIRE(nverts < 0x7fffffffll && ntris < 0x2aaaaaaall, IVX_EINVAL, "ca_smoothing: mesh too large for 32-bit ids");
    if (nverts == 0) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    if (vdtype == IVX_F32)
        return run_ca<float>((float *)verts, nverts, faces, ntris, normals, t, tmax, bmin, n_iters, nullptr, staircase_out,
                             weights_out, true, st);
    return run_ca<double>((double *)verts, nverts, faces, ntris, normals, t, tmax, bmin, n_iters, nullptr, staircase_out,
                          weights_out, true, st);
}

extern "C" int ivx_dev_mesh_propagate_weights(const void *verts, int vdtype, int64_t nverts, const int32_t *faces,
                                              int64_t ntris, const uint8_t *seed_flags, double tmax, double bmin,
                                              double *weights, void *stream) {
    IVX_REQUIRE(vdtype == IVX_F32 || vdtype == IVX_F64, IVX_EINVAL, "propagate_weights: vertices must be float32 or float64");
    IVX_REQUIRE(nverts >= 0 && ntris >= 0, IVX_EINVAL, "propagate_weights: negative size");
    IVX_REQUIRE(nverts < 0x7fffffffll && ntris < 0x2aaaaaaall, IVX_EINVAL, "propagate_weights: mesh too large");
    if (nverts == 0) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    if (vdtype == IVX_F32)
        return run_ca<float>((float *)verts, nverts, faces, ntris, nullptr, 0.0, tmax, bmin, 0, seed_flags, nullptr, weights,
                             false, st);
    return run_ca<double>((double *)verts, nverts, faces, ntris, nullptr, 0.0, tmax, bmin, 0, seed_flags, nullptr, weights,
                          false, st);
}

extern "C" int ivx_dev_mesh_face_normals(const void *verts, int vdtype, const int32_t *faces, int64_t ntris, double *normals,
                                         void *stream) {
    IVX_REQUIRE(vdtype == IVX_F32 || vdtype == IVX_F64, IVX_EINVAL, "face_normals: vertices must be float32 or float64");
    if (ntris <= 0) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    if (vdtype == IVX_F32)
        hipLaunchKernelGGL((k_face_normals<float>), dim3(grid_for(ntris)), dim3(256), 0, st, (const float *)verts, faces, ntris,
                           normals);
    else
        hipLaunchKernelGGL((k_face_normals<double>), dim3(grid_for(ntris)), dim3(256), 0, st, (const double *)verts, faces,
                           ntris, normals);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// ---- host forms -----------------------------------------------------------------------------------------------------
namespace {
static int check_faces(const int32_t *faces, int64_t ntris, int64_t nverts) {
    for (int64_t q = 0; q < 3 * ntris; q++)
        IVX_REQUIRE(faces[q] >= 0 && faces[q] < nverts, IVX_EDOM, "mesh: face index %d outside [0, %lld)", faces[q],
                    (long long)nverts);
    return IVX_OK;
}
} // namespace

// mode 0: full smoothing (normals required unless NULL -> computed from the geometry); mode 1: weights only from seed_flags
extern "C" int ivx_context_aware_smoothing(void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                                           const double *normals, double t, double tmax, double bmin, int n_iters,
                                           uint8_t *staircase_out, double *weights_out) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(vdtype == IVX_F32 || vdtype == IVX_F64, IVX_EINVAL, "ca_smoothing: vertices must be float32 or float64");
    IVX_REQUIRE(nverts >= 0 && ntris >= 0 && n_iters >= 0, IVX_EINVAL, "ca_smoothing: negative size");
    if (nverts == 0) return IVX_OK;
    int rc;
    if ((rc = check_faces(faces, ntris, nverts))) return rc;
    const size_t vs = vdtype == IVX_F32 ? 4 : 8;
    void *d_v, *d_f, *d_n, *d_s, *d_w;
    if ((rc = ws_get(WS_IN, (size_t)nverts * 3 * vs + 16, &d_v))) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)ntris * 12 + 16, &d_f))) return rc;
    if ((rc = ws_get(WS_AUX1, (size_t)ntris * 24 + 16, &d_n))) return rc;
    if ((rc = ws_get(WS_AUX2, (size_t)nverts + 16, &d_s))) return rc;
    if ((rc = ws_get(WS_AUX3, (size_t)nverts * 8 + 16, &d_w))) return rc;
    IVX_HIP(hipMemcpy(d_v, verts, (size_t)nverts * 3 * vs, hipMemcpyHostToDevice));
    if (ntris) IVX_HIP(hipMemcpy(d_f, faces, (size_t)ntris * 12, hipMemcpyHostToDevice));
    if (normals) {
        if (ntris) IVX_HIP(hipMemcpy(d_n, normals, (size_t)ntris * 24, hipMemcpyHostToDevice));
    } else if ((rc = ivx_dev_mesh_face_normals(d_v, vdtype, (const int32_t *)d_f, ntris, (double *)d_n, nullptr)))
        return rc;
    if ((rc = ivx_dev_context_aware_smoothing(d_v, vdtype, nverts, (const int32_t *)d_f, ntris, (const double *)d_n, t, tmax,
                                              bmin, n_iters, (uint8_t *)d_s, (double *)d_w, nullptr)))
        return rc;
    IVX_HIP(hipDeviceSynchronize());
    IVX_HIP(hipMemcpy(verts, d_v, (size_t)nverts * 3 * vs, hipMemcpyDeviceToHost));
    if (staircase_out) IVX_HIP(hipMemcpy(staircase_out, d_s, (size_t)nverts, hipMemcpyDeviceToHost));
    if (weights_out) IVX_HIP(hipMemcpy(weights_out, d_w, (size_t)nverts * 8, hipMemcpyDeviceToHost));
    return IVX_OK;
}

extern "C" int ivx_mesh_propagate_weights(const void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                                          const uint8_t *seed_flags, double tmax, double bmin, double *weights) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(vdtype == IVX_F32 || vdtype == IVX_F64, IVX_EINVAL, "propagate_weights: vertices must be float32 or float64");
    IVX_REQUIRE(nverts >= 0 && ntris >= 0, IVX_EINVAL, "propagate_weights: negative size");
    if (nverts == 0) return IVX_OK;
    int rc;
    if ((rc = check_faces(faces, ntris, nverts))) return rc;
    const size_t vs = vdtype == IVX_F32 ? 4 : 8;
    void *d_v, *d_f, *d_s, *d_w;
    if ((rc = ws_get(WS_IN, (size_t)nverts * 3 * vs + 16, &d_v))) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)ntris * 12 + 16, &d_f))) return rc;
    if ((rc = ws_get(WS_AUX2, (size_t)nverts + 16, &d_s))) return rc;
    if ((rc = ws_get(WS_AUX3, (size_t)nverts * 8 + 16, &d_w))) return rc;
    IVX_HIP(hipMemcpy(d_v, verts, (size_t)nverts * 3 * vs, hipMemcpyHostToDevice));
    if (ntris) IVX_HIP(hipMemcpy(d_f, faces, (size_t)ntris * 12, hipMemcpyHostToDevice));
    IVX_HIP(hipMemcpy(d_s, seed_flags, (size_t)nverts, hipMemcpyHostToDevice));
    if ((rc = ivx_dev_mesh_propagate_weights(d_v, vdtype, nverts, (const int32_t *)d_f, ntris, (const uint8_t *)d_s, tmax, bmin,
                                             (double *)d_w, nullptr)))
        return rc;
    IVX_HIP(hipDeviceSynchronize());
    IVX_HIP(hipMemcpy(weights, d_w, (size_t)nverts * 8, hipMemcpyDeviceToHost));
    return IVX_OK;
}

extern "C" int ivx_mesh_face_normals(const void *verts, int vdtype, int64_t nverts, const int32_t *faces, int64_t ntris,
                                     double *normals) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(vdtype == IVX_F32 || vdtype == IVX_F64, IVX_EINVAL, "face_normals: vertices must be float32 or float64");
    if (ntris <= 0) return IVX_OK;
    int rc;
    if ((rc = check_faces(faces, ntris, nverts))) return rc;
    const size_t vs = vdtype == IVX_F32 ? 4 : 8;
    void *d_v, *d_f, *d_n;
    if ((rc = ws_get(WS_IN, (size_t)nverts * 3 * vs + 16, &d_v))) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)ntris * 12 + 16, &d_f))) return rc;
    if ((rc = ws_get(WS_AUX1, (size_t)ntris * 24 + 16, &d_n))) return rc;
    IVX_HIP(hipMemcpy(d_v, verts, (size_t)nverts * 3 * vs, hipMemcpyHostToDevice));
    IVX_HIP(hipMemcpy(d_f, faces, (size_t)ntris * 12, hipMemcpyHostToDevice));
    if ((rc = ivx_dev_mesh_face_normals(d_v, vdtype, (const int32_t *)d_f, ntris, (double *)d_n, nullptr))) return rc;
    IVX_HIP(hipMemcpy(normals, d_n, (size_t)ntris * 24, hipMemcpyDeviceToHost));
    return IVX_OK;
}
