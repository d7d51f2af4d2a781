// k_mc.hip -- marching cubes over a (virtually) padded + Y-flipped piece: the geometry of
// create_surface_piece (invesalius/data/surface_process.py:52-68,100-186; converters.py:34-101), whose
// contouring step in the reference is vtkContourFilter (VTK 9.3, third party).  Case table: the generated
// include/ivx_mc_tables.h shared with the CPU oracle (oracle/ivx_oracle.c orc_marching_cubes).
//
// MI355X design (memory-bound, no MFMA):
//   1. k_mc_bits     one streaming pass over the voxels (2 B/voxel int16, 1 B/voxel uint8; 16-B loads per
//                    lane) -> "inside" bit planes (scalar >= iso), 1 bit per padded grid point, both
//                    iso-values in the same pass.  A 512^3 piece gives a 19 MB bit volume per iso: it lives in
//                    L2 / Infinity Cache for the rest of the pipeline.
//   2. k_mc_count    one lane per 64-cell word: four row words (+ the carry bit of the next word) give the
//                    active-cell mask with a handful of 64-bit ops; only active cells look up the case table.
//                    Per-word triangle counts (u16) + per-workgroup sums.
//   3. k_mc_scan     exclusive scan of the per-workgroup sums (u64 offsets), single workgroup.
//   4. k_mc_emit     same traversal; wave-level prefix (DPP shuffles) + LDS across the 4 waves gives every word
//                    its output slot; active cells gather their 8 scalars and write 9 x f32 per triangle.
// Output order == the oracle's (iso-major, then k, j, i raster order of cells), so parity is an array compare.
// Vertex arithmetic is done in double and rounded once to float32, exactly like the oracle.
#include "ivx_internal.h"

#define MC_TABLE_QUAL __device__ const
#include "../../include/ivx_mc_tables.h"

typedef short short8_t __attribute__((ext_vector_type(8)));
typedef unsigned char uchar8_t __attribute__((ext_vector_type(8)));

namespace {

struct Geom {
    int64_t nz, ny, nx;  // piece
    int64_t NZ, NY, NX;  // padded grid points
    int64_t WX;          // uint64 words per padded point row
    int64_t WC;          // uint64 words per cell row  (NX-1 cells)
    int64_t nrows;       // (NZ-1)*(NY-1) cell rows
    int pxy, pb;
    double padv;
    double sx, sy, sz;
    int64_t yoff, zoff;
};

static int make_geom(const ivx_mc_params *p, Geom *g) {
    IVX_REQUIRE(p && p->nz >= 0 && p->ny >= 0 && p->nx >= 0, IVX_EINVAL, "mc: bad shape");
    IVX_REQUIRE(p->niso >= 1 && p->niso <= 2, IVX_EINVAL, "mc: niso must be 1 or 2");
    IVX_REQUIRE(p->dtype == IVX_U8 || p->dtype == IVX_I16 || p->dtype == IVX_U16, IVX_EINVAL, "mc: dtype");
    g->nz = p->nz; g->ny = p->ny; g->nx = p->nx;
    g->pxy = p->pad_xy ? 1 : 0; g->pb = p->pad_bottom ? 1 : 0;
    g->NZ = p->nz + g->pb + (p->pad_top ? 1 : 0);
    g->NY = p->ny + 2 * g->pxy; g->NX = p->nx + 2 * g->pxy;
    g->WX = ivx::cdiv(g->NX, 64);
    g->WC = g->NX > 1 ? ivx::cdiv(g->NX - 1, 64) : 0;
    g->nrows = (g->NZ > 1 && g->NY > 1) ? (g->NZ - 1) * (g->NY - 1) : 0;
    g->padv = p->pad_value;
    g->sx = p->spacing[0]; g->sy = p->spacing[1]; g->sz = p->spacing[2];
    g->yoff = g->NY - 1 - g->pxy;
    g->zoff = p->roi_start - p->vtk_pz;
    return IVX_OK;
}

// scratch layout (all 256-B aligned): bits[niso][NZ*NY*WX] u64 | counts[niso][nwords] u16 |
// blocksum[niso*nblocks] u32 | blockoff[niso*nblocks+1] u64
struct Scratch {
    size_t bits_words, nwords, nblocks;
    size_t off_bits, off_counts, off_bsum, off_boff, total;
};
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static Scratch make_scratch(const Geom &g, int niso) {
    Scratch s;
    s.bits_words = (size_t)(g.NZ * g.NY * g.WX);
    s.nwords = (size_t)(g.nrows * g.WC);
    s.nblocks = (s.nwords + 255) / 256;
    s.off_bits = 0;
    s.off_counts = al256(s.off_bits + (size_t)niso * s.bits_words * 8);
    s.off_bsum = al256(s.off_counts + (size_t)niso * s.nwords * 2);
    s.off_boff = al256(s.off_bsum + (size_t)niso * s.nblocks * 4);
    s.total = al256(s.off_boff + ((size_t)niso * s.nblocks + 1) * 8);
    return s;
}

template <typename T>
__device__ __forceinline__ double mc_at(const T *a, const Geom &g, int64_t k, int64_t jf, int64_t i) {
    const int64_t ja = (g.NY - 1 - jf) - g.pxy, ia = i - g.pxy, ka = k - g.pb;
    if (ia < 0 || ia >= g.nx || ja < 0 || ja >= g.ny || ka < 0 || ka >= g.nz) return g.padv;
    return (double)a[(ka * g.ny + ja) * g.nx + ia];
}

// ---- 1. inside-bit planes -----------------------------------------------------------------------
// one lane per output BYTE (8 padded grid points of one row); byte b of word w <-> points 64w+8b .. +7
template <typename T, int NISO>
__global__ __launch_bounds__(256) void k_mc_bits(const T *__restrict__ a, Geom g, double iso0, double iso1,
                                                 uint8_t *__restrict__ bits0, uint8_t *__restrict__ bits1) {
    const int64_t bytes_per_row = g.WX * 8;
    const int64_t total = g.NZ * g.NY * bytes_per_row;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec_rows = (g.nx % 8 == 0) && (((uintptr_t)a & 15) == 0);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t row = t / bytes_per_row;
        const int64_t q = t - row * bytes_per_row;
        const int64_t k = row / g.NY, jf = row - k * g.NY;
        const int64_t ja = (g.NY - 1 - jf) - g.pxy, ka = k - g.pb;
        const int64_t x0 = q * 8; // first padded point of this byte
        unsigned m0 = 0, m1 = 0;
        if (x0 < g.NX) {
            const bool row_in = ja >= 0 && ja < g.ny && ka >= 0 && ka < g.nz;
            double v[8];
            const int64_t s0 = x0 - g.pxy; // source x of point 0
            if (!row_in) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = g.padv;
            } else {
                const T *r = a + (ka * g.ny + ja) * g.nx;
                if (sizeof(T) == 2 && vec_rows && x0 + 8 <= g.nx) {
                    // aligned 16-B chunk [x0, x0+8) of the source row; with pxy the byte covers source
                    // [x0-1, x0+7): element -1 comes from a 2-B load (same cache line as the neighbour lane's chunk)
                    const short8_t c = *reinterpret_cast<const short8_t *>(r + x0);
                    if (g.pxy) {
                        v[0] = x0 > 0 ? (double)r[x0 - 1] : g.padv;
#pragma unroll
                        for (int e = 1; e < 8; e++) v[e] = (double)(T)c[e - 1];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = (double)(T)c[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int64_t sx = s0 + e;
                        v[e] = (sx >= 0 && sx < g.nx) ? (double)r[sx] : g.padv;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const bool valid = x0 + e < g.NX;
                m0 |= (valid && v[e] >= iso0) ? (1u << e) : 0u;
                if (NISO == 2) m1 |= (valid && v[e] >= iso1) ? (1u << e) : 0u;
            }
        }
        bits0[t] = (uint8_t)m0;
        if (NISO == 2) bits1[t] = (uint8_t)m1;
    }
}

// ---- shared traversal: the 8 corner-bit words of the 64 cells of (row, word w) ----------------------
struct Corner8 {
    uint64_t c[8];
    uint64_t active;
};
__device__ __forceinline__ Corner8 load_corners(const uint64_t *__restrict__ bits, const Geom &g, int64_t k,
                                                int64_t j, int64_t w) {
    Corner8 r;
    const bool has_next = (w + 1) < g.WX;
#pragma unroll
    for (int dz = 0; dz < 2; dz++)
#pragma unroll
        for (int dy = 0; dy < 2; dy++) {
            const uint64_t *row = bits + ((k + dz) * g.NY + (j + dy)) * g.WX;
            const uint64_t lo = row[w];
            const uint64_t nx = has_next ? row[w + 1] : 0ull;
            r.c[4 * dz + 2 * dy] = lo;
            r.c[4 * dz + 2 * dy + 1] = (lo >> 1) | (nx << 63);
        }
    uint64_t any = 0, all = ~0ull;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        any |= r.c[c];
        all &= r.c[c];
    }
    const int64_t ncell = (g.NX - 1) - w * 64; // cells of this word that exist
    const uint64_t valid = ncell >= 64 ? ~0ull : ((1ull << ncell) - 1ull);
    r.active = any & ~all & valid;
    return r;
}
__device__ __forceinline__ int case_of(const Corner8 &r, int b) {
    int idx = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) idx |= (int)((r.c[c] >> b) & 1ull) << c;
    return idx;
}

// ---- 2. count -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mc_count(const uint64_t *__restrict__ bits, Geom g, size_t nwords,
                                                  uint16_t *__restrict__ counts, uint32_t *__restrict__ bsum) {
    __shared__ uint32_t s_part[4];
    const size_t wid = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n = 0;
    if (wid < nwords) {
        const int64_t row = (int64_t)(wid / (size_t)g.WC), w = (int64_t)(wid - (size_t)row * g.WC);
        const int64_t k = row / (g.NY - 1), j = row - k * (g.NY - 1);
        const Corner8 r = load_corners(bits, g, k, j, w);
        uint64_t act = r.active;
        while (act) {
            const int b = __builtin_ctzll(act);
            act &= act - 1;
            n += MC_NTRI[case_of(r, b)];
        }
        counts[wid] = (uint16_t)n;
    }
    // workgroup sum: wave reduce + 4-entry LDS
    uint32_t s = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// ---- 3. scan of workgroup sums (single workgroup of 1024) ---------------------------------------------
__global__ __launch_bounds__(1024) void k_mc_scan(const uint32_t *__restrict__ bsum, size_t n,
                                                  uint64_t *__restrict__ boff) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t base = 0; base < n; base += 1024) {
        const size_t i = base + threadIdx.x;
        const uint64_t v = i < n ? (uint64_t)bsum[i] : 0ull;
        uint64_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        uint64_t wbase = 0;
        for (int q = 0; q < wv; q++) wbase += s_wave[q];
        const uint64_t carry = s_carry;
        if (i < n) boff[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wbase + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) boff[n] = s_carry;
}

// ---- 4. emit ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_mc_emit(const T *__restrict__ a, const uint64_t *__restrict__ bits, Geom g,
                                                 size_t nwords, double iso, const uint16_t *__restrict__ counts,
                                                 const uint64_t *__restrict__ boff, uint64_t out_base,
                                                 float *__restrict__ tris, uint64_t max_tris) {
    __shared__ uint32_t s_wave[4];
    const size_t wid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = wid < nwords ? (uint32_t)counts[wid] : 0u;
    uint32_t inc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int q = 0; q < wv; q++) wbase += s_wave[q];
    if (n == 0) return;
    uint64_t off = out_base + boff[blockIdx.x] + wbase + (inc - n);
    if (off + n > max_tris) return; // caller sized the buffer from the count; never write past it

    const int64_t row = (int64_t)(wid / (size_t)g.WC), w = (int64_t)(wid - (size_t)row * g.WC);
    const int64_t k = row / (g.NY - 1), j = row - k * (g.NY - 1);
    const Corner8 r = load_corners(bits, g, k, j, w);
    uint64_t act = r.active;
    while (act) {
        const int b = __builtin_ctzll(act);
        act &= act - 1;
        const int idx = case_of(r, b);
        const int nt = MC_NTRI[idx];
        if (!nt) continue;
        const int64_t i = w * 64 + b;
        double sc[8];
#pragma unroll
        for (int c = 0; c < 8; c++) sc[c] = mc_at(a, g, k + ((c >> 2) & 1), j + ((c >> 1) & 1), i + (c & 1));
        float *o = tris + off * 9;
        for (int t = 0; t < 3 * nt; t++) {
            const int e = MC_TRI[idx][t];
            const int c0 = MC_EDGE_CORNERS[e][0], c1 = MC_EDGE_CORNERS[e][1];
            const double tt = (iso - sc[c0]) / (sc[c1] - sc[c0]);
            double p0 = (double)(i + MC_EDGE_BASE[e][0] - g.pxy);
            double p1 = (double)(j + MC_EDGE_BASE[e][1] - g.yoff);
            double p2 = (double)(k + MC_EDGE_BASE[e][2] + g.zoff);
            const int ax = MC_EDGE_AXIS[e];
            if (ax == 0) p0 += tt;
            else if (ax == 1) p1 += tt;
            else p2 += tt;
            o[3 * t + 0] = (float)(g.sx * p0);
            o[3 * t + 1] = (float)(g.sy * p1);
            o[3 * t + 2] = (float)(g.sz * p2);
        }
        off += nt;
    }
}

template <typename T>
static int run_bits(const ivx_mc_params *p, const Geom &g, const Scratch &s, const void *a, char *scratch,
                    hipStream_t st) {
    const int64_t total = g.NZ * g.NY * g.WX * 8;
    if (total == 0) return IVX_OK;
    const int64_t blocks = ivx::cdiv(total, 256);
    const int grid = (int)(blocks < 32768 ? blocks : 32768);
    uint8_t *b0 = (uint8_t *)(scratch + s.off_bits);
    uint8_t *b1 = b0 + s.bits_words * 8;
    if (p->niso == 2)
        hipLaunchKernelGGL((k_mc_bits<T, 2>), dim3(grid), dim3(256), 0, st, (const T *)a, g, p->iso[0], p->iso[1], b0,
                           b1);
    else
        hipLaunchKernelGGL((k_mc_bits<T, 1>), dim3(grid), dim3(256), 0, st, (const T *)a, g, p->iso[0], 0.0, b0, b0);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

template <typename T>
static int run_emit(const ivx_mc_params *p, const Geom &g, const Scratch &s, const void *a, const char *scratch,
                    float *tris, int64_t max_tris, hipStream_t st) {
    if (s.nblocks == 0) return IVX_OK;
    const uint64_t *boff = (const uint64_t *)(scratch + s.off_boff);
    // iso 1's output starts where iso 0's ends: boff is one scan over [iso0 blocks | iso1 blocks]
    for (int q = 0; q < p->niso; q++) {
        const uint64_t *bits = (const uint64_t *)(scratch + s.off_bits) + (size_t)q * s.bits_words;
        const uint16_t *counts = (const uint16_t *)(scratch + s.off_counts) + (size_t)q * s.nwords;
        hipLaunchKernelGGL((k_mc_emit<T>), dim3((unsigned)s.nblocks), dim3(256), 0, st, (const T *)a, bits, g,
                           s.nwords, p->iso[q], counts, boff + (size_t)q * s.nblocks, (uint64_t)0, tris,
                           (uint64_t)max_tris);
        IVX_LAUNCH_CHECK();
    }
    return IVX_OK;
}

} // namespace

extern "C" int ivx_dev_mc_scratch_bytes(const ivx_mc_params *p, size_t *nbytes) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    *nbytes = make_scratch(g, p->niso).total;
    return IVX_OK;
}

extern "C" int ivx_dev_mc_count(const ivx_mc_params *p, const void *a, void *scratch_, int64_t *ntris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    const Scratch s = make_scratch(g, p->niso);
    char *scratch = (char *)scratch_;
    hipStream_t st = ivx::S(stream);
    *ntris = 0;
    if (s.nwords == 0) return IVX_OK;
    IVX_REQUIRE(s.nblocks * (size_t)p->niso < 0x7fffffffull, IVX_EINVAL, "mc: piece too large for one launch");
    switch (p->dtype) {
    case IVX_U8: rc = run_bits<uint8_t>(p, g, s, a, scratch, st); break;
    case IVX_I16: rc = run_bits<int16_t>(p, g, s, a, scratch, st); break;
    default: rc = run_bits<uint16_t>(p, g, s, a, scratch, st); break;
    }
    if (rc) return rc;
    uint32_t *bsum = (uint32_t *)(scratch + s.off_bsum);
    uint64_t *boff = (uint64_t *)(scratch + s.off_boff);
    for (int q = 0; q < p->niso; q++) {
        const uint64_t *bits = (const uint64_t *)(scratch + s.off_bits) + (size_t)q * s.bits_words;
        uint16_t *counts = (uint16_t *)(scratch + s.off_counts) + (size_t)q * s.nwords;
        hipLaunchKernelGGL(k_mc_count, dim3((unsigned)s.nblocks), dim3(256), 0, st, bits, g, s.nwords, counts,
                           bsum + (size_t)q * s.nblocks);
        IVX_LAUNCH_CHECK();
    }
    const size_t nb = s.nblocks * (size_t)p->niso;
    hipLaunchKernelGGL(k_mc_scan, dim3(1), dim3(1024), 0, st, bsum, nb, boff);
    IVX_LAUNCH_CHECK();
    uint64_t total = 0;
    IVX_HIP(hipMemcpyAsync(&total, boff + nb, 8, hipMemcpyDeviceToHost, st));
    IVX_HIP(hipStreamSynchronize(st));
    *ntris = (int64_t)total;
    return IVX_OK;
}

extern "C" int ivx_dev_mc_emit(const ivx_mc_params *p, const void *a, const void *scratch, float *tris,
                               int64_t max_tris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    const Scratch s = make_scratch(g, p->niso);
    if (s.nwords == 0 || max_tris <= 0) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    switch (p->dtype) {
    case IVX_U8: return run_emit<uint8_t>(p, g, s, a, (const char *)scratch, tris, max_tris, st);
    case IVX_I16: return run_emit<int16_t>(p, g, s, a, (const char *)scratch, tris, max_tris, st);
    default: return run_emit<uint16_t>(p, g, s, a, (const char *)scratch, tris, max_tris, st);
    }
}

extern "C" int ivx_marching_cubes(const ivx_mc_params *p, const void *a, const int64_t strides[3], float *tris,
                                  int64_t max_tris, int64_t *ntris) {
    using namespace ivx;
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    const size_t isz = dtype_size(p->dtype);
    const int64_t shape[3] = {p->nz, p->ny, p->nx};
    const size_t n = (size_t)p->nz * p->ny * p->nx;
    void *d_a, *d_scr;
    size_t sb;
    if ((rc = ivx_dev_mc_scratch_bytes(p, &sb))) return rc;
    if ((rc = ws_get(WS_IN, n * isz, &d_a))) return rc;
    if ((rc = ws_get(WS_AUX0, sb, &d_scr))) return rc;
    if ((rc = upload_strided(d_a, a, shape, strides, isz, WS_IN))) return rc;
    int64_t cnt = 0;
    if ((rc = ivx_dev_mc_count(p, d_a, d_scr, &cnt, nullptr))) return rc;
    *ntris = cnt;
    if (!tris || cnt == 0) return IVX_OK;
    IVX_REQUIRE(max_tris >= cnt, IVX_ERANGE, "mc: output buffer holds %lld triangles, %lld needed", (long long)max_tris,
                (long long)cnt);
    void *d_tris;
    if ((rc = ws_get(WS_OUT, (size_t)cnt * 36, &d_tris))) return rc;
    if ((rc = ivx_dev_mc_emit(p, d_a, d_scr, (float *)d_tris, cnt, nullptr))) return rc;
    IVX_HIP(hipMemcpy(tris, d_tris, (size_t)cnt * 36, hipMemcpyDeviceToHost));
    return IVX_OK;
}
