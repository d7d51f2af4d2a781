// k_mesh.hip -- what join_process_surface (invesalius/data/surface_process.py:204-472) does to the joined surface
// after the point merge, for the two stages that are plain scans over the mesh:
//   * "keep largest"      vtkPolyDataConnectivityFilter, SetExtractionModeToLargestRegion  (surface_process.py:376-391)
//   * area / volume       vtkMassProperties                                                (surface_process.py:452-458)
// Both VTK classes are third party and absent from the reference tree; the algorithms restated here are the
// published ones (connected regions over shared points, largest by CELL count, first region wins a tie;
// Alyassin et al. 1994 discrete-divergence volume with max-unit-normal-component weighting + Heron areas).
//
// MI355X design: both are HBM/L2-bound integer+double streams, no MFMA.
//   components   lock-free union-find over VERTEX ids (atomicMin links only -- plain stores into parent[] lose
//                links, see k_ccl.hip), one lane per triangle; roots flattened once; triangles per root counted with
//                workgroup-level pre-aggregation of the dominant root (otherwise ~every triangle of a real surface
//                would hit ONE address); compaction = two exclusive scans (faces kept, vertices used).
//   mass         one lane per triangle, double accumulators, fixed-shape tree reduction (wave shuffle -> LDS ->
//                per-workgroup partials -> one workgroup) so the result is reproducible run to run.
#include <algorithm>

#include "ivx_internal.h"
#include "scan_u32.h"

namespace {

__device__ __forceinline__ uint32_t uf_find(const uint32_t *parent, uint32_t x) {
    for (;;) {
        const uint32_t p = parent[x];
        if (p == x) return x;
        x = p;
    }
}
__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; } // the larger root goes under the smaller
        const uint32_t old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_mesh_init(uint32_t *__restrict__ parent, uint32_t *__restrict__ cnt,
                                                   uint32_t *__restrict__ mintri, uint32_t *__restrict__ usedv, int64_t nv,
                                                   unsigned long long *__restrict__ best, uint32_t *__restrict__ nreg) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v <= nv; v += stride) {
        if (v < nv) {
            parent[v] = (uint32_t)v;
            cnt[v] = 0u;
            mintri[v] = 0xffffffffu;
        }
        usedv[v] = 0u; // nv + 1 entries: the scan leaves the total in the last one
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *best = 0ull;
        *nreg = 0u;
    }
}

__global__ __launch_bounds__(256) void k_mesh_union(const int32_t *__restrict__ faces, int64_t nt,
                                                    uint32_t *__restrict__ parent) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const uint32_t a = (uint32_t)faces[3 * t], b = (uint32_t)faces[3 * t + 1], c = (uint32_t)faces[3 * t + 2];
    uf_union(parent, a, b);
    uf_union(parent, a, c);
}

__global__ __launch_bounds__(256) void k_mesh_flatten(const uint32_t *__restrict__ parent, uint32_t *__restrict__ root,
                                                      int64_t nv) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride)
        root[v] = uf_find(parent, (uint32_t)v);
}

// triangles per region + the smallest triangle id of each region (the tie-break: VTK numbers regions by their first
// cell and keeps the FIRST region among equally large ones)
constexpr int TPL = 8; // triangles per lane
__global__ __launch_bounds__(256) void k_mesh_tricount(const int32_t *__restrict__ faces, int64_t nt,
                                                       const uint32_t *__restrict__ root, uint32_t *__restrict__ cnt,
                                                       uint32_t *__restrict__ mintri) {
    __shared__ uint32_t s_hot, s_hot_cnt, s_hot_min;
    const int64_t base = (int64_t)blockIdx.x * 256 * TPL;
    if (threadIdx.x == 0) {
        s_hot = base < nt ? root[(uint32_t)faces[3 * base]] : 0xffffffffu;
        s_hot_cnt = 0u;
        s_hot_min = 0xffffffffu;
    }
    __syncthreads();
    const uint32_t hot = s_hot;
    uint32_t my_hot = 0, my_min = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < TPL; q++) {
        const int64_t t = base + (int64_t)q * 256 + threadIdx.x;
        const bool ok = t < nt;
        const uint32_t r = ok ? root[(uint32_t)faces[3 * t]] : 0xffffffffu;
        if (ok && r == hot) {
            my_hot++;
            my_min = my_min < (uint32_t)t ? my_min : (uint32_t)t;
        }
        // everything else: one atomic per distinct root per wave
        bool todo = ok && r != hot;
        while (__ballot(todo)) {
            const unsigned long long pending = __ballot(todo);
            const int leader = __builtin_ctzll(pending);
            const uint32_t lr = __shfl(r, leader, 64);
            const bool mine = todo && r == lr;
            const unsigned long long grp = __ballot(mine);
            if ((int)(threadIdx.x & 63) == leader) {
                atomicAdd(&cnt[lr], (uint32_t)__popcll(grp));
                atomicMin(&mintri[lr], (uint32_t)(base + (int64_t)q * 256 + (threadIdx.x & ~63) + __builtin_ctzll(grp)));
            }
            todo = todo && !mine;
        }
    }
    // workgroup total for the hot root
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        my_hot += __shfl_xor(my_hot, o, 64);
        const uint32_t m = __shfl_xor(my_min, o, 64);
        my_min = my_min < m ? my_min : m;
    }
    if ((threadIdx.x & 63) == 0 && my_hot) {
        atomicAdd(&s_hot_cnt, my_hot);
        atomicMin(&s_hot_min, my_min);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_hot_cnt) {
        atomicAdd(&cnt[hot], s_hot_cnt);
        atomicMin(&mintri[hot], s_hot_min);
    }
}

// best = max over regions of (cells << 32) | ~first_cell  ->  most cells, then the earliest first cell
__global__ __launch_bounds__(256) void k_mesh_best(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ mintri,
                                                   int64_t nv, unsigned long long *__restrict__ best,
                                                   uint32_t *__restrict__ nreg) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long key = 0ull;
    uint32_t regions = 0;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
        const uint32_t c = cnt[v];
        if (c) {
            regions++;
            const unsigned long long k = ((unsigned long long)c << 32) | (unsigned long long)(~mintri[v]);
            key = k > key ? k : key;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long k = __shfl_xor(key, o, 64);
        key = k > key ? k : key;
        regions += __shfl_xor(regions, o, 64);
    }
    // one atomic pair per WORKGROUP (the grid is capped at 1024 of them): ~50 k same-address atomics took 1 ms
    __shared__ unsigned long long s_key[4];
    __shared__ uint32_t s_reg[4];
    if ((threadIdx.x & 63) == 0) {
        s_key[threadIdx.x >> 6] = key;
        s_reg[threadIdx.x >> 6] = regions;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; q++) {
            key = s_key[q] > key ? s_key[q] : key;
            regions += s_reg[q];
        }
        if (key) atomicMax(best, key);
        if (regions) atomicAdd(nreg, regions);
    }
}

// keep[t] = 1 for the triangles of the winning region (recognised by its first cell); used[v] = 1 for their vertices
__global__ __launch_bounds__(256) void k_mesh_mark(const int32_t *__restrict__ faces, int64_t nt,
                                                   const uint32_t *__restrict__ root, const uint32_t *__restrict__ mintri,
                                                   const unsigned long long *__restrict__ best, uint32_t *__restrict__ keep,
                                                   uint32_t *__restrict__ usedv) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nt) return;
    if (t == nt) {
        keep[t] = 0u;
        return;
    }
    const uint32_t first = ~(uint32_t)(*best & 0xffffffffull);
    const uint32_t a = (uint32_t)faces[3 * t], b = (uint32_t)faces[3 * t + 1], c = (uint32_t)faces[3 * t + 2];
    const bool k = mintri[root[a]] == first;
    keep[t] = k ? 1u : 0u;
    if (k) {
        usedv[a] = 1u;
        usedv[b] = 1u;
        usedv[c] = 1u;
    }
}

__global__ __launch_bounds__(256) void k_mesh_compact_faces(const int32_t *__restrict__ faces, int64_t nt,
                                                            const uint32_t *__restrict__ koff,
                                                            const uint32_t *__restrict__ voff, int32_t *__restrict__ out,
                                                            int64_t max_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const uint32_t o = koff[t];
    if (koff[t + 1] == o || (int64_t)o >= max_out) return;
#pragma unroll
    for (int q = 0; q < 3; q++) out[3 * (int64_t)o + q] = (int32_t)voff[(uint32_t)faces[3 * t + q]];
}

__global__ __launch_bounds__(256) void k_mesh_compact_verts(const float *__restrict__ verts, int64_t nv,
                                                            const uint32_t *__restrict__ voff, float *__restrict__ out,
                                                            int64_t max_out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const uint32_t o = voff[v];
    if (voff[v + 1] == o || (int64_t)o >= max_out) return;
#pragma unroll
    for (int q = 0; q < 3; q++) out[3 * (int64_t)o + q] = verts[3 * v + q];
}

// ---- mass properties ------------------------------------------------------------------------------------------------
// partial record: area, vol x/y/z, then the seven counters munc x/y/z, wxyz, wxy, wxz, wyz (as doubles: exact < 2^53)
constexpr int NP = 11;
__device__ __forceinline__ void tri_mass(const float *__restrict__ verts, const int32_t *__restrict__ faces, int64_t t,
                                         double acc[NP]) {
    double x[3], y[3], z[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int64_t v = faces ? (int64_t)faces[3 * t + q] : 3 * t + q;
        x[q] = (double)verts[3 * v];
        y[q] = (double)verts[3 * v + 1];
        z[q] = (double)verts[3 * v + 2];
    }
    const double i0 = x[1] - x[0], j0 = y[1] - y[0], k0 = z[1] - z[0];
    const double i1 = x[2] - x[0], j1 = y[2] - y[0], k1 = z[2] - z[0];
    const double i2 = x[2] - x[1], j2 = y[2] - y[1], k2 = z[2] - z[1];
    double u0 = j0 * k1 - k0 * j1, u1 = k0 * i1 - i0 * k1, u2 = i0 * j1 - j0 * i1;
    const double len = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
    if (len != 0.0) {
        u0 /= len;
        u1 /= len;
        u2 /= len;
    } else {
        u0 = u1 = u2 = 0.0;
    }
    const double a0 = fabs(u0), a1 = fabs(u1), a2 = fabs(u2);
    int slot = -1;
    if (a0 > a1 && a0 > a2) slot = 4;
    else if (a1 > a0 && a1 > a2) slot = 5;
    else if (a2 > a0 && a2 > a1) slot = 6;
    else if (a0 == a1 && a0 == a2) slot = 7;
    else if (a0 == a1 && a0 > a2) slot = 8;
    else if (a0 == a2 && a0 > a1) slot = 9;
    else if (a1 == a2 && a0 < a2) slot = 10;
    const double a = sqrt(i1 * i1 + j1 * j1 + k1 * k1);
    const double b = sqrt(i0 * i0 + j0 * j0 + k0 * k0);
    const double c = sqrt(i2 * i2 + j2 * j2 + k2 * k2);
    const double s = 0.5 * (a + b + c);
    const double area = sqrt(fabs(s * (s - a) * (s - b) * (s - c)));
    acc[0] += area;
    acc[1] += area * u0 * ((x[0] + x[1] + x[2]) / 3.0);
    acc[2] += area * u1 * ((y[0] + y[1] + y[2]) / 3.0);
    acc[3] += area * u2 * ((z[0] + z[1] + z[2]) / 3.0);
#pragma unroll
    for (int q = 4; q < NP; q++) acc[q] += (slot == q) ? 1.0 : 0.0;
}

__device__ __forceinline__ void block_reduce_np(double acc[NP], double *__restrict__ out) {
    __shared__ double s_part[4][NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_xor(acc[q], o, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < NP; q++) s_part[wv][q] = acc[q];
    __syncthreads();
    if (threadIdx.x < NP) out[threadIdx.x] = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

constexpr int MASS_TPL = 4;
__global__ __launch_bounds__(256) void k_mesh_mass(const float *__restrict__ verts, const int32_t *__restrict__ faces,
                                                   int64_t nt, double *__restrict__ partial) {
    double acc[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) acc[q] = 0.0;
    const int64_t base = (int64_t)blockIdx.x * 256 * MASS_TPL;
#pragma unroll
    for (int q = 0; q < MASS_TPL; q++) {
        const int64_t t = base + (int64_t)q * 256 + threadIdx.x;
        if (t < nt) tri_mass(verts, faces, t, acc);
    }
    block_reduce_np(acc, partial + (int64_t)blockIdx.x * NP);
}
__global__ __launch_bounds__(256) void k_mesh_mass_final(const double *__restrict__ partial, int64_t nb, double ncells,
                                                         double *__restrict__ out) {
    __shared__ double s_tot[NP];
    double acc[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) acc[q] = 0.0;
    for (int64_t b = threadIdx.x; b < nb; b += 256)
#pragma unroll
        for (int q = 0; q < NP; q++) acc[q] += partial[b * NP + q];
    block_reduce_np(acc, s_tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double *m = s_tot;
        const double kx = (m[4] + m[7] / 3.0 + (m[8] + m[9]) / 2.0) / ncells;
        const double ky = (m[5] + m[7] / 3.0 + (m[8] + m[10]) / 2.0) / ncells;
        const double kz = (m[6] + m[7] / 3.0 + (m[9] + m[10]) / 2.0) / ncells;
        out[0] = fabs(kx * m[1] + ky * m[2] + kz * m[3]);
        out[1] = m[0];
        out[2] = m[1];
        out[3] = m[2];
        out[4] = m[3];
        out[5] = kx;
        out[6] = ky;
        out[7] = kz;
    }
}

struct MeshLayout { // per-stream WS_MESH
    size_t off_parent, off_root, off_cnt, off_min, off_keep, off_used, off_bsum, off_misc, total;
};
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static MeshLayout mesh_layout(int64_t nv, int64_t nt) {
    MeshLayout m;
    const size_t v4 = al256(((size_t)nv + 1) * 4), t4 = al256(((size_t)nt + 1) * 4);
    m.off_parent = 0;
    m.off_root = m.off_parent + v4;
    m.off_cnt = m.off_root + v4;
    m.off_min = m.off_cnt + v4;
    m.off_used = m.off_min + v4;
    m.off_keep = m.off_used + v4;
    m.off_bsum = m.off_keep + t4;
    const int64_t nmax = nv > nt ? nv : nt;
    m.off_misc = m.off_bsum + al256(((size_t)scan_u32_blocks(nmax + 1) + 16) * 4);
    m.total = m.off_misc + 256;
    return m;
}

} // namespace

extern "C" int ivx_dev_mesh_keep_largest(const float *verts, int64_t nverts, const int32_t *faces, int64_t ntris,
                                         float *out_verts, int64_t max_verts, int32_t *out_faces, int64_t max_tris,
                                         int64_t *out_nverts, int64_t *out_ntris, int64_t *nregions, void *stream) {
    IVX_REQUIRE(nverts >= 0 && ntris >= 0, IVX_EINVAL, "mesh: negative size");
    IVX_REQUIRE(nverts < 0x7fffffffll && ntris < 0x7fffffffll, IVX_EINVAL, "mesh: more than 2^31 vertices / triangles");
    *out_nverts = 0;
    *out_ntris = 0;
    if (nregions) *nregions = 0;
    if (ntris == 0) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    const MeshLayout m = mesh_layout(nverts, ntris);
    void *ws;
    int rc;
    if ((rc = ivx::ws_get_s(ivx::WS_MESH, st, m.total, &ws))) return rc;
    char *w = (char *)ws;
    uint32_t *parent = (uint32_t *)(w + m.off_parent), *root = (uint32_t *)(w + m.off_root);
    uint32_t *cnt = (uint32_t *)(w + m.off_cnt), *mintri = (uint32_t *)(w + m.off_min);
    uint32_t *usedv = (uint32_t *)(w + m.off_used), *keep = (uint32_t *)(w + m.off_keep);
    uint32_t *bsum = (uint32_t *)(w + m.off_bsum);
    unsigned long long *best = (unsigned long long *)(w + m.off_misc);
    uint32_t *nreg = (uint32_t *)(w + m.off_misc + 8), *tot_t = nreg + 1, *tot_v = nreg + 2;
    const unsigned gv = (unsigned)std::min<int64_t>(ivx::cdiv(nverts + 1, 256), 16384);
    const unsigned gt = (unsigned)ivx::cdiv(ntris + 1, 256);
    hipLaunchKernelGGL(k_mesh_init, dim3(gv), dim3(256), 0, st, parent, cnt, mintri, usedv, nverts, best, nreg);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_union, dim3(gt), dim3(256), 0, st, faces, ntris, parent);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_flatten, dim3(gv), dim3(256), 0, st, parent, root, nverts);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_tricount, dim3((unsigned)ivx::cdiv(ntris, 256 * TPL)), dim3(256), 0, st, faces, ntris, root, cnt,
                       mintri);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_best, dim3(std::min(gv, 1024u)), dim3(256), 0, st, cnt, mintri, nverts, best, nreg);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_mark, dim3(gt), dim3(256), 0, st, faces, ntris, root, mintri, best, keep, usedv);
    IVX_LAUNCH_CHECK();
    if ((rc = scan_u32_exclusive(keep, ntris + 1, bsum, tot_t, st))) return rc;
    if ((rc = scan_u32_exclusive(usedv, nverts + 1, bsum, tot_v, st))) return rc;
    uint32_t seq, got[3];
    if ((rc = ivx::mailbox_publish(nreg, 3, st, &seq))) return rc;
    if ((rc = ivx::mailbox_wait(seq, st, got, 3))) return rc;
    if (nregions) *nregions = got[0];
    *out_ntris = got[1];
    *out_nverts = got[2];
    if (!out_verts || !out_faces) return IVX_OK;
    IVX_REQUIRE(max_tris >= (int64_t)got[1] && max_verts >= (int64_t)got[2], IVX_ERANGE,
                "mesh: output buffers too small (%u verts, %u triangles needed)", got[2], got[1]);
    hipLaunchKernelGGL(k_mesh_compact_faces, dim3((unsigned)ivx::cdiv(ntris, 256)), dim3(256), 0, st, faces, ntris, keep, usedv,
                       out_faces, max_tris);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_compact_verts, dim3((unsigned)ivx::cdiv(nverts, 256)), dim3(256), 0, st, verts, nverts, usedv,
                       out_verts, max_verts);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_mesh_mass_properties(const float *verts, const int32_t *faces, int64_t ntris, double *out8,
                                            void *stream) {
    IVX_REQUIRE(ntris >= 0, IVX_EINVAL, "mesh: negative size");
    hipStream_t st = ivx::S(stream);
    if (ntris == 0) {
        IVX_HIP(hipMemsetAsync(out8, 0, 8 * sizeof(double), st));
        return IVX_OK;
    }
    const int64_t nb = ivx::cdiv(ntris, 256 * MASS_TPL);
    void *ws;
    int rc;
    if ((rc = ivx::ws_get_s(ivx::WS_MESH2, st, (size_t)nb * NP * 8, &ws))) return rc;
    hipLaunchKernelGGL(k_mesh_mass, dim3((unsigned)nb), dim3(256), 0, st, verts, faces, ntris, (double *)ws);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_mass_final, dim3(1), dim3(256), 0, st, (const double *)ws, nb, (double)ntris, out8);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// ---- host forms -----------------------------------------------------------------------------------------------------
extern "C" int ivx_mesh_keep_largest(const float *verts, int64_t nverts, const int32_t *faces, int64_t ntris,
                                     float *out_verts, int32_t *out_faces, int64_t *out_nverts, int64_t *out_ntris,
                                     int64_t *nregions) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(nverts >= 0 && ntris >= 0, IVX_EINVAL, "mesh: negative size");
    *out_nverts = *out_ntris = 0;
    if (nregions) *nregions = 0;
    if (ntris == 0) return IVX_OK;
    for (int64_t q = 0; q < 3 * ntris; q++)
        IVX_REQUIRE(faces[q] >= 0 && faces[q] < nverts, IVX_EDOM, "mesh: face index %d outside [0, %lld)", faces[q],
                    (long long)nverts);
    void *d_v, *d_f, *d_ov, *d_of;
    int rc;
    if ((rc = ws_get(WS_IN, (size_t)nverts * 12 + 16, &d_v))) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)ntris * 12 + 16, &d_f))) return rc;
    if ((rc = ws_get(WS_OUT, (size_t)nverts * 12 + 16, &d_ov))) return rc;
    if ((rc = ws_get(WS_AUX1, (size_t)ntris * 12 + 16, &d_of))) return rc;
    IVX_HIP(hipMemcpy(d_v, verts, (size_t)nverts * 12, hipMemcpyHostToDevice));
    IVX_HIP(hipMemcpy(d_f, faces, (size_t)ntris * 12, hipMemcpyHostToDevice));
    if ((rc = ivx_dev_mesh_keep_largest((const float *)d_v, nverts, (const int32_t *)d_f, ntris, (float *)d_ov, nverts,
                                        (int32_t *)d_of, ntris, out_nverts, out_ntris, nregions, nullptr)))
        return rc;
    IVX_HIP(hipDeviceSynchronize());
    if (out_verts && *out_nverts) IVX_HIP(hipMemcpy(out_verts, d_ov, (size_t)*out_nverts * 12, hipMemcpyDeviceToHost));
    if (out_faces && *out_ntris) IVX_HIP(hipMemcpy(out_faces, d_of, (size_t)*out_ntris * 12, hipMemcpyDeviceToHost));
    return IVX_OK;
}

extern "C" int ivx_mesh_mass_properties(const float *verts, int64_t nverts, const int32_t *faces, int64_t ntris,
                                        double *out8) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(nverts >= 0 && ntris >= 0, IVX_EINVAL, "mesh: negative size");
    for (int q = 0; q < 8; q++) out8[q] = 0.0;
    if (ntris == 0) return IVX_OK;
    if (faces) {
        for (int64_t q = 0; q < 3 * ntris; q++)
            IVX_REQUIRE(faces[q] >= 0 && faces[q] < nverts, IVX_EDOM, "mesh: face index %d outside [0, %lld)", faces[q],
                        (long long)nverts);
    } else {
        IVX_REQUIRE(nverts == 3 * ntris, IVX_EINVAL, "mesh: a soup needs 3 vertices per triangle");
    }
    void *d_v, *d_f = nullptr, *d_o;
    int rc;
    if ((rc = ws_get(WS_IN, (size_t)nverts * 12 + 16, &d_v))) return rc;
    if (faces && (rc = ws_get(WS_AUX0, (size_t)ntris * 12 + 16, &d_f))) return rc;
    if ((rc = ws_get(WS_SMALL, 64, &d_o))) return rc;
    IVX_HIP(hipMemcpy(d_v, verts, (size_t)nverts * 12, hipMemcpyHostToDevice));
    if (faces) IVX_HIP(hipMemcpy(d_f, faces, (size_t)ntris * 12, hipMemcpyHostToDevice));
    if ((rc = ivx_dev_mesh_mass_properties((const float *)d_v, (const int32_t *)d_f, ntris, (double *)d_o, nullptr))) return rc;
    IVX_HIP(hipMemcpy(out8, d_o, 64, hipMemcpyDeviceToHost));
    return IVX_OK;
}
