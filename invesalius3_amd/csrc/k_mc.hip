// k_mc.hip -- marching cubes over a (virtually) padded + Y-flipped piece: the geometry of
// create_surface_piece (invesalius/data/surface_process.py:52-68,100-186; converters.py:34-101), whose
// contouring step in the reference is vtkContourFilter (VTK 9.3, third party; on vtkImageData it delegates to
// vtkSynchronizedTemplates3D: own templates table, point-merged output in its own order).  Case table here: the generated
// include/ivx_mc_tables.h shared with the CPU oracle (oracle/ivx_oracle.c orc_marching_cubes) -- the vertex set and the
// closedness of the surface are pinned (table-independent), per-cell triangulation and triangle order vs VTK are not.
//
// MI355X design (memory-bound, no MFMA):
//   1. k_mc_bits     one streaming pass over the voxels (2 B/voxel int16, 1 B/voxel uint8; one aligned 16-B load
//                    per lane) -> "inside" bit planes (scalar >= iso), 1 bit per SOURCE voxel (64 voxels of an
//                    x-row per uint64, the same layout the region-growing kernels use), both iso-values in the
//                    same pass.  A 512^3 piece gives a 16 MiB bit volume per iso: it lives in L2 / Infinity
//                    Cache for the rest of the pipeline.  The one-voxel padding and the Y flip are applied
//                    when the words are read (funnel shift by one bit, constant pad rows), never stored.
//   2. k_mc_count    one lane per 64-cell word: four row words (+ the carry bit of the next word) give the
//                    active-cell mask with a handful of 64-bit ops; only active cells look up the case table.
//                    Per-word triangle counts (u16) + per-workgroup sums.
//   3. k_mc_scan     exclusive scan of the per-workgroup sums (u64 offsets), one workgroup per 16 384 sums.
//   4. k_mc_emit     one workgroup per 256 words.  Corner words + a block scan of the counts go to LDS; then the
//                    workgroup walks its TRIANGLES 256 at a time, one lane per triangle (binary search of the
//                    owning word in LDS, short walk over that word's active cells), so lanes stay busy however
//                    unevenly the surface is spread.  The 9 floats of each triangle are staged in LDS
//                    (stride 9 dwords: conflict-free) and leave as fully coalesced dword stores: HBM sees each
//                    output line once.
// Output order == the oracle's (iso-major, then k, j, i raster order of cells), so parity is an array compare.
// Vertex arithmetic is done in double and rounded once to float32, exactly like the oracle.
#include <cmath>
#include <map>
#include <mutex>

#include "ivx_internal.h"
#include "scan_u32.h"

#define MC_TABLE_QUAL __device__ __attribute__((aligned(16))) const
#include "../../include/ivx_mc_tables.h"

typedef short short8_t __attribute__((ext_vector_type(8)));
typedef unsigned char uchar16_t __attribute__((ext_vector_type(16)));

namespace {

struct Geom {
    int64_t nz, ny, nx;  // piece
    int64_t NZ, NY, NX;  // padded grid points
    int64_t ws;          // uint64 words per SOURCE row = ceil(nx/64)
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
    g->ws = ivx::cdiv(g->nx, 64);
    g->WX = ivx::cdiv(g->NX, 64);
    g->WC = g->NX > 1 ? ivx::cdiv(g->NX - 1, 64) : 0;
    g->nrows = (g->NZ > 1 && g->NY > 1) ? (g->NZ - 1) * (g->NY - 1) : 0;
    // (the triangle list names a cell word by 16-bit slice, 16-bit row and 15-bit word-in-row)
    IVX_REQUIRE(g->NZ <= 65536 && g->NY <= 65536 && g->WC <= 32768, IVX_EINVAL, "mc: piece too large (at most 65 535 cells along z and y, 2^21 along x)");
    g->padv = p->pad_value;
    g->sx = p->spacing[0]; g->sy = p->spacing[1]; g->sz = p->spacing[2];
    g->yoff = g->NY - 1 - g->pxy;
    g->zoff = p->roi_start - p->vtk_pz;
    return IVX_OK;
}

// scratch layout (all 256-B aligned): bits[niso][nz*ny*ws] u64 | counts[niso][nwords] u16 |
// blocksum[niso*nblocks] u32 | blockoff[niso*nblocks+1] u64
struct Scratch {
    size_t bits_words, nwords, nblocks;
    size_t off_bits, off_counts, off_bsum, off_boff, total;
};
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static Scratch make_scratch(const Geom &g, int niso) {
    Scratch s;
    s.bits_words = (size_t)(g.nz * g.ny * g.ws);
    s.nwords = (size_t)(g.nrows * g.WC);
    s.nblocks = (s.nwords + 255) / 256;
    s.off_bits = 0;
    s.off_counts = al256(s.off_bits + (size_t)niso * s.bits_words * 8 + 16);
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

// ---- 1. inside-bit planes in SOURCE coordinates -------------------------------------------------------
// One lane per aligned 16-byte chunk of a source row: 8 voxels (2-byte types) -> 1 output byte, 16 voxels
// (uint8) -> 2 output bytes.  Byte b of word w <-> voxels 64w+8b .. +7; bits beyond nx stay 0.
template <typename T, int NISO>
__global__ __launch_bounds__(256) void k_mc_bits(const T *__restrict__ a, int64_t nrows_src, int64_t nx, int64_t ws,
                                                 double iso0, double iso1, uint8_t *__restrict__ bits0,
                                                 uint8_t *__restrict__ bits1) {
    constexpr int V = 16 / sizeof(T);           // voxels per lane
    constexpr int OB = V / 8;                   // output bytes per lane
    const int64_t cpr = ws * 64 / V;            // chunks per (word-padded) row
    const int64_t total = nrows_src * cpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = (nx % V == 0) && (((uintptr_t)a & 15) == 0);
    if (vec && ws * 64 == nx) {
        // rows are whole words: the volume is one flat array of chunks; 4 independent 16-B loads per lane in flight
        const int64_t nchunk = nrows_src * (nx / V);
        typedef T vec_t __attribute__((ext_vector_type(V)));
        const vec_t *src = reinterpret_cast<const vec_t *>(a);
        for (int64_t t0 = ((int64_t)blockIdx.x * blockDim.x) * 4 + threadIdx.x; t0 < nchunk; t0 += stride * 4) {
            vec_t x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t t = t0 + (int64_t)u * blockDim.x;
                if (t < nchunk) x[u] = __builtin_nontemporal_load(&src[t]);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t t = t0 + (int64_t)u * blockDim.x;
                if (t >= nchunk) continue;
                unsigned m0 = 0, m1 = 0;
#pragma unroll
                for (int e = 0; e < V; e++) {
                    const double d = (double)(T)x[u][e];
                    m0 |= d >= iso0 ? (1u << e) : 0u;
                    if (NISO == 2) m1 |= d >= iso1 ? (1u << e) : 0u;
                }
                if (OB == 1) {
                    bits0[t] = (uint8_t)m0;
                    if (NISO == 2) bits1[t] = (uint8_t)m1;
                } else {
                    reinterpret_cast<uint16_t *>(bits0)[t] = (uint16_t)m0;
                    if (NISO == 2) reinterpret_cast<uint16_t *>(bits1)[t] = (uint16_t)m1;
                }
            }
        }
        return;
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t row = t / cpr, q = t - row * cpr;
        const int64_t x0 = q * V;
        unsigned m0 = 0, m1 = 0;
        if (x0 < nx) {
            const T *r = a + row * nx + x0;
            T v[V];
            if (vec) { // x0 + V <= nx because nx % V == 0
                if (sizeof(T) == 2) {
                    const short8_t c = *reinterpret_cast<const short8_t *>(r);
#pragma unroll
                    for (int e = 0; e < V; e++) v[e] = (T)c[e % 8];
                } else {
                    const uchar16_t c = *reinterpret_cast<const uchar16_t *>(r);
#pragma unroll
                    for (int e = 0; e < V; e++) v[e] = (T)c[e % 16];
                }
            } else {
#pragma unroll
                for (int e = 0; e < V; e++) v[e] = (x0 + e < nx) ? r[e] : (T)0;
            }
#pragma unroll
            for (int e = 0; e < V; e++) {
                const bool valid = vec || (x0 + e < nx);
                const double d = (double)v[e];
                m0 |= (valid && d >= iso0) ? (1u << e) : 0u;
                if (NISO == 2) m1 |= (valid && d >= iso1) ? (1u << e) : 0u;
            }
        }
        const int64_t ob = (row * ws * 8) + q * OB;
        if (OB == 1) {
            bits0[ob] = (uint8_t)m0;
            if (NISO == 2) bits1[ob] = (uint8_t)m1;
        } else {
            *reinterpret_cast<uint16_t *>(bits0 + ob) = (uint16_t)m0;
            if (NISO == 2) *reinterpret_cast<uint16_t *>(bits1 + ob) = (uint16_t)m1;
        }
    }
}

// ---- padded + flipped view of the source bit planes -----------------------------------------------------
// Words w and w+1 of padded point row (k, jf): padded x = 64w .. 64w+127.  Pad rows / pad columns carry pbits.
// Written without branches around the loads: the three source words are always fetched (clamped, always-valid
// addresses) and masked afterwards, so the 12 loads of a cell word are all in flight together.
__device__ __forceinline__ void padded_pair(const uint64_t *__restrict__ S, const Geom &g, int64_t k, int64_t jf,
                                            int64_t w, uint64_t pbits, uint64_t &lo, uint64_t &hi) {
    const int64_t ja = (g.NY - 1 - jf) - g.pxy, ka = k - g.pb;
    const bool row_in = ja >= 0 && ja < g.ny && ka >= 0 && ka < g.nz;
    // clamp into [0, n-1], and to 0 when the piece is EMPTY along that axis (n == 0: every row is padding, the loads
    // then hit word 0 of the scratch block, which always exists, and are masked away)
    const auto clampi = [](int64_t v, int64_t n) { return v >= n ? (n > 0 ? n - 1 : 0) : (v < 0 ? 0 : v); };
    const int64_t rj = clampi(ja, g.ny), rk = clampi(ka, g.nz);
    const uint64_t *row = S + (rk * g.ny + rj) * g.ws;
    // source words w-1, w, w+1 (clamped index, masked when outside [0, ws) or when the row is padding)
    const int64_t wm = clampi(w - 1, g.ws), wc = clampi(w, g.ws), wp = clampi(w + 1, g.ws);
    uint64_t sm = row[wm], sc = row[wc], sp = row[wp];
    sm = (row_in && w - 1 >= 0 && w - 1 < g.ws) ? sm : 0ull;
    sc = (row_in && w < g.ws) ? sc : 0ull;
    sp = (row_in && w + 1 < g.ws) ? sp : 0ull;
    // bits of source x in [64w - pxy, 64w + 64 - pxy) and the following 64
    uint64_t v0 = g.pxy ? ((sc << 1) | (sm >> 63)) : sc;
    uint64_t v1 = g.pxy ? ((sp << 1) | (sc >> 63)) : sp;
    // positions of each word that exist in the padded row, and those backed by source voxels
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int64_t ww = w + q;
        const int64_t rem = g.NX - ww * 64; // padded points in this word
        const uint64_t exist = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
        uint64_t src = row_in ? exist : 0ull;
        if (g.pxy && ww == 0) src &= ~1ull;
        const int64_t top = g.pxy + g.nx - ww * 64; // first padded-x (relative) beyond the source
        src &= top >= 64 ? ~0ull : (top <= 0 ? 0ull : ((1ull << top) - 1ull));
        uint64_t &v = q == 0 ? v0 : v1;
        v = (v & src) | (pbits & exist & ~src);
    }
    lo = v0;
    hi = v1;
}

struct Corner8 {
    uint64_t c[8];
    uint64_t active;
};
// The eight corner words of cell word (k, j, w).  Same view as padded_pair, specialised: of the following word only
// bit 0 is ever needed (corner x+1 of cell 63), so each of the four rows costs two loads instead of three and one mask
// instead of two, and the masks that depend only on w are computed once for all four rows.
__device__ __forceinline__ Corner8 load_corners(const uint64_t *__restrict__ bits, const Geom &g, int64_t k,
                                                int64_t j, int64_t w, uint64_t pbits) {
    Corner8 r;
    const auto clampi = [](int64_t v, int64_t n) { return v >= n ? (n > 0 ? n - 1 : 0) : (v < 0 ? 0 : v); };
    const int64_t rem = g.NX - w * 64; // padded points in this word
    const uint64_t exist = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
    uint64_t srcm = exist; // ... of which backed by source voxels (when the row is a source row)
    if (g.pxy && w == 0) srcm &= ~1ull;
    const int64_t top = g.pxy + g.nx - w * 64;
    srcm &= top >= 64 ? ~0ull : (top <= 0 ? 0ull : ((1ull << top) - 1ull));
    const int64_t x1 = (w + 1) * 64;              // padded x of the next word's first point
    const bool exist1 = x1 < g.NX;
    const bool src1 = x1 - g.pxy < g.nx;          // (x1 - pxy >= 0 always)
    const bool has_m = w - 1 >= 0 && w - 1 < g.ws, has_c = w < g.ws, has_p = w + 1 < g.ws;
    const int64_t wm = clampi(w - 1, g.ws), wc = clampi(w, g.ws), wp = clampi(w + 1, g.ws);
    uint64_t a0[4], a1[4];
    bool rin[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { // issue the eight loads back to back
        const int64_t ja = (g.NY - 1 - (j + (q & 1))) - g.pxy, ka = k + (q >> 1) - g.pb;
        rin[q] = ja >= 0 && ja < g.ny && ka >= 0 && ka < g.nz;
        const uint64_t *row = bits + (clampi(ka, g.nz) * g.ny + clampi(ja, g.ny)) * g.ws;
        a0[q] = row[wc];
        a1[q] = g.pxy ? row[wm] : row[wp];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint64_t sc = (rin[q] && has_c) ? a0[q] : 0ull;
        uint64_t v0, raw1;
        if (g.pxy) {
            const uint64_t sm = (rin[q] && has_m) ? a1[q] : 0ull;
            v0 = (sc << 1) | (sm >> 63);
            raw1 = sc >> 63;
        } else {
            const uint64_t sp = (rin[q] && has_p) ? a1[q] : 0ull;
            v0 = sc;
            raw1 = sp & 1ull;
        }
        const uint64_t src = rin[q] ? srcm : 0ull;
        const uint64_t lo = (v0 & src) | (pbits & exist & ~src);
        const uint64_t nb = exist1 ? ((rin[q] && src1) ? raw1 : (pbits & 1ull)) : 0ull;
        r.c[2 * q] = lo;
        r.c[2 * q + 1] = (lo >> 1) | (nb << 63);
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
__device__ __forceinline__ int case_of(const uint64_t *c, int b) {
    int idx = 0;
    if (b != 63) {
        // c[2q+1] is c[2q] moved down by one cell, so bits b and b+1 of the even words ARE the corner pairs (2q, 2q+1)
#pragma unroll
        for (int q = 0; q < 4; q++) idx |= (int)((c[2 * q] >> b) & 3ull) << (2 * q);
    } else {
#pragma unroll
        for (int q = 0; q < 8; q++) idx |= (int)((c[q] >> 63) & 1ull) << q;
    }
    return idx;
}

// ---- 2. count -----------------------------------------------------------------------------------
// Two phases per workgroup.  (a) every lane loads the eight corner words of its cell word and finds the active cells: most
// words have none (13 % do on the bench surface).  (b) the words that have some are handed, packed, to the first lanes of
// the workgroup, which walk their cells (one table look-up per active cell): the serial walks of a workgroup then sit in
// ONE or two full waves instead of being scattered over four mostly idle ones (the kernel is bound by the walks, not by
// the loads: 37 -> 2x us before / after on the bench volume).
__global__ __launch_bounds__(256) void k_mc_count(const uint64_t *__restrict__ bits, Geom g, size_t nwords,
                                                  uint64_t pbits, uint16_t *__restrict__ counts,
                                                  uint32_t *__restrict__ bsum) {
    __shared__ uint32_t s_part[4], s_wcnt[4];
    __shared__ uint8_t s_ntri[256];
    __shared__ uint64_t s_c[256][5]; // per active word: the four even corner words + the active mask
    __shared__ uint8_t s_hi[256];    // bit 63 of the four odd corner words (cell 63's far corners)
    __shared__ uint16_t s_slot[256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    s_ntri[tid] = MC_NTRI[tid];
    const size_t wid = (size_t)blockIdx.x * 256 + tid;
    uint64_t act = 0;
    Corner8 r;
    if (wid < nwords) {
        // nwords < 2^32 (checked on the host): 32-bit divisions instead of two 64-bit ones per lane
        const uint32_t row = (uint32_t)wid / (uint32_t)g.WC, w = (uint32_t)wid - row * (uint32_t)g.WC;
        const uint32_t k = row / (uint32_t)(g.NY - 1), j = row - k * (uint32_t)(g.NY - 1);
        r = load_corners(bits, g, (int64_t)k, (int64_t)j, (int64_t)w, pbits);
        act = r.active;
        if (!act) counts[wid] = 0;
    }
    const unsigned long long am = __ballot(act != 0);
    if (lane == 0) s_wcnt[wv] = (uint32_t)__popcll(am);
    __syncthreads();
    uint32_t before = 0;
    for (int q = 0; q < wv; q++) before += s_wcnt[q];
    const uint32_t nact = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    if (act) {
        const uint32_t slot = before + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
        s_slot[slot] = (uint16_t)tid;
        uint32_t hi = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            s_c[slot][q] = r.c[2 * q];
            hi |= (uint32_t)(r.c[2 * q + 1] >> 63) << q;
        }
        s_c[slot][4] = act;
        s_hi[slot] = (uint8_t)hi;
    }
    __syncthreads();
    uint32_t n = 0;
    if ((uint32_t)tid < nact) {
        const uint64_t c0 = s_c[tid][0], c1 = s_c[tid][1], c2 = s_c[tid][2], c3 = s_c[tid][3];
        uint64_t a2 = s_c[tid][4];
        const uint32_t hi = s_hi[tid];
        while (a2) {
            const int b = __builtin_ctzll(a2);
            a2 &= a2 - 1;
            int idx;
            if (b != 63)
                idx = (int)((c0 >> b) & 3ull) | ((int)((c1 >> b) & 3ull) << 2) | ((int)((c2 >> b) & 3ull) << 4) | ((int)((c3 >> b) & 3ull) << 6);
            else
                idx = (int)(c0 >> 63) | ((int)(hi & 1u) << 1) | ((int)(c1 >> 63) << 2) | ((int)(hi >> 1 & 1u) << 3) | ((int)(c2 >> 63) << 4) |
                      ((int)(hi >> 2 & 1u) << 5) | ((int)(c3 >> 63) << 6) | ((int)(hi >> 3 & 1u) << 7);
            n += s_ntri[idx];
        }
        counts[(size_t)blockIdx.x * 256 + s_slot[tid]] = (uint16_t)n;
    }
    uint32_t s = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) s_part[wv] = s;
    __syncthreads();
    if (tid == 0) bsum[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// ---- 3. scan of workgroup sums (single workgroup of 1024) ---------------------------------------------
// (Tried in round 3: the scan in k_mc_count's tail, done by the last workgroup to sign a ticket -- one launch less, but the
// 9 259 same-address agent-scope atomics of the ticket serialise at ~8 ns each: mc_count 44 -> 124 us.  Dropped.)
__global__ __launch_bounds__(1024) void k_mc_scan(const uint32_t *__restrict__ bsum, size_t n,
                                                  uint64_t *__restrict__ boff) {
    // One workgroup per 16 384 sums (one at 512^3, five at 1024^3 -- where the single workgroup of rounds 1 - 5 took 68 us, 5 % of
    // the stage).  No hand-over between workgroups: each first adds up everything in front of its chunk itself (a coalesced
    // read of at most a few hundred KB out of the L2) and then scans its own chunk.
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    constexpr int PER = 16; // consecutive elements per lane
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * (1024 * PER);
    uint64_t pre = 0;
    { // (base is a multiple of 16 384: whole 16-byte groups, four independent loads in flight per lane and trip)
        const uint4 *b4 = reinterpret_cast<const uint4 *>(bsum);
        const size_t n4 = base / 4;
        uint64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        size_t i = threadIdx.x;
        for (; i + 3 * 1024 < n4; i += 4 * 1024) {
            const uint4 a = b4[i], b = b4[i + 1024], c = b4[i + 2048], d = b4[i + 3072];
            p0 += (uint64_t)a.x + a.y + a.z + a.w;
            p1 += (uint64_t)b.x + b.y + b.z + b.w;
            p2 += (uint64_t)c.x + c.y + c.z + c.w;
            p3 += (uint64_t)d.x + d.y + d.z + d.w;
        }
        for (; i < n4; i += 1024) {
            const uint4 a = b4[i];
            p0 += (uint64_t)a.x + a.y + a.z + a.w;
        }
        pre = p0 + p1 + p2 + p3;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
    if (lane == 0) s_wave[wv] = pre;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t c = 0;
        for (int q = 0; q < 16; q++) c += s_wave[q];
        s_carry = c;
    }
    __syncthreads();
    {
        const size_t i0 = base + (size_t)threadIdx.x * PER;
        uint32_t v[PER];
        uint64_t sum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            v[q] = i0 + q < n ? bsum[i0 + q] : 0u;
            sum += v[q];
        }
        uint64_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        __syncthreads(); // (s_wave is reused)
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        uint64_t off = s_carry + inc - sum;
        for (int q = 0; q < wv; q++) off += s_wave[q];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            if (i0 + q < n) boff[i0 + q] = off;
            off += v[q];
        }
        if (threadIdx.x == 1023 && base + 1024 * PER >= n) boff[n] = off; // the last chunk's last lane holds the total
    }
}

// ---- 4. emit: one lane per TRIANGLE ---------------------------------------------------------------
// edge e -> (axis, low corner, low-corner offset) without tables: edges 0-3 run along x with (dy,dz) = (e&1, e>>1&1),
// 4-7 along y with (dx,dz), 8-11 along z with (dx,dy)  (tools/gen_mc_tables.py conventions).
__device__ __forceinline__ void edge_decode(int e, int &ax, int &bx, int &by, int &bz) {
    ax = e >> 2;
    const int a = e & 1, b = (e >> 1) & 1;
    bx = ax == 0 ? 0 : a;
    by = ax == 0 ? a : (ax == 1 ? 0 : b);
    bz = ax == 2 ? 0 : b;
}

// 4a. list: the OWNER of each cell word writes one 64-bit descriptor per triangle into a flat global list, in output
//     order: (k << 48) | (j << 32) | (w << 17) | (cell bit << 11) | (case << 3) | triangle-in-case, (k, j, w) = the cell
//     word's slice, row and word in the row (the readers -- one lane per triangle -- then need no divisions: two 32-bit
//     divisions by run-time divisors were a quarter of k_mc_emit's instructions).  One case evaluation per
//     active cell; only workgroups that own triangles do anything beyond a 256-entry scan.  Like the count, in two
//     phases: the words that own triangles are handed, packed, to the first lanes of the workgroup, so that the corner
//     loads and the serial walks fill one or two waves instead of idling in four.
__global__ __launch_bounds__(256) void k_mc_list(const uint64_t *__restrict__ bits, Geom g, size_t nwords, uint64_t pbits,
                                                 const uint16_t *__restrict__ counts, const uint64_t *__restrict__ boff,
                                                 uint64_t *__restrict__ list, uint64_t max_tris) {
    __shared__ uint32_t s_wave[4], s_wcnt[4];
    __shared__ uint8_t s_ntri[256];
    __shared__ uint16_t s_slot[256];
    __shared__ uint32_t s_pos[256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    s_ntri[tid] = MC_NTRI[tid];
    const size_t wid0 = (size_t)blockIdx.x * 256;
    const uint32_t n = wid0 + tid < nwords ? (uint32_t)counts[wid0 + tid] : 0u;
    uint32_t inc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const unsigned long long am = __ballot(n != 0);
    if (lane == 63) s_wave[wv] = inc;
    if (lane == 0) s_wcnt[wv] = (uint32_t)__popcll(am);
    __syncthreads();
    const uint32_t nact = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    if (!nact) return; // (uniform)
    if (n) {
        uint32_t wbase = 0, before = 0;
        for (int q = 0; q < wv; q++) {
            wbase += s_wave[q];
            before += s_wcnt[q];
        }
        const uint32_t slot = before + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
        s_slot[slot] = (uint16_t)tid;
        s_pos[slot] = wbase + inc - n; // first triangle of the word, relative to the block
    }
    __syncthreads();
    if ((uint32_t)tid >= nact) return;
    const size_t wid = wid0 + s_slot[tid];
    uint64_t pos = boff[blockIdx.x] + s_pos[tid];
    const uint32_t nmine = (uint32_t)counts[wid];
    if (pos + nmine > max_tris) return; // never write past the list the caller sized from the count
    const uint32_t row = (uint32_t)wid / (uint32_t)g.WC, w = (uint32_t)wid - row * (uint32_t)g.WC;
    const uint32_t k = row / (uint32_t)(g.NY - 1), j = row - k * (uint32_t)(g.NY - 1);
    const Corner8 r = load_corners(bits, g, (int64_t)k, (int64_t)j, (int64_t)w, pbits);
    uint64_t act = r.active;
    while (act) {
        const int b = __builtin_ctzll(act);
        act &= act - 1;
        const int idx = case_of(r.c, b);
        const uint32_t nt = s_ntri[idx];
        const uint64_t d0 = ((uint64_t)k << 48) | ((uint64_t)j << 32) | ((uint64_t)w << 17) | ((uint64_t)b << 11) | ((uint64_t)idx << 3);
#pragma unroll
        for (uint32_t t = 0; t < MC_MAX_TRI; t++)
            if (t < nt) list[pos + t] = d0 | t;
        pos += nt;
    }
}

// 4b. emit: a flat, regular kernel -- one lane per triangle of the list, 256 consecutive triangles per workgroup.
//     Nothing but the 9-KB staging buffer in LDS, so eight workgroups share a CU and hide each other's gather latency.
// LEVELS: the voxel values are not gathered but derived -- `a` is a mask known to hold v_out outside the inside plane, v_sel
// where `sel` has a bit and v_in elsewhere inside (a resident pipeline's threshold + region-growing result).  Which end of
// an edge is inside is in the case index already, so a triangle costs three bit look-ups in a 16 MiB plane instead of six
// byte gathers from the mask; the interpolation then runs on the same numbers and gives the same bits.
struct McLevels {
    const uint64_t *sel; // source-coordinate plane, rows of g.ws words
    double v_out, v_in, v_sel;
    // (iso - s0) / (s1 - s0) for the four (which end is inside, which inside level) combinations, divided once on the host:
    // IEEE double division gives the same bits there as three divisions per triangle give here
    double tt[4]; // [in0 * 2 + sel]
};
static inline McLevels make_levels(const uint64_t *sel, double iso, double v_out, double v_in, double v_sel) {
    McLevels lv{sel, v_out, v_in, v_sel, {0.0, 0.0, 0.0, 0.0}};
    for (int in0 = 0; in0 < 2; in0++)
        for (int q = 0; q < 2; q++) {
            const double vin = q ? v_sel : v_in, s0 = in0 ? vin : v_out, s1 = in0 ? v_out : vin;
            lv.tt[in0 * 2 + q] = (iso - s0) / (s1 - s0);
        }
    return lv;
}
// The nine floats of ONE triangle: cell (k, j, i) of the padded grid, case `idx`, triangle `rel` of the case.  s_tri = the
// triangle table (rows of 15), s_e0 / s_e1 / s_ec = the per-edge constants (see k_mc_emit).  Shared by the list-driven emit
// and the single-pass surface kernel: same arithmetic, same bits.
template <typename T, bool LEVELS>
__device__ __forceinline__ void mc_triangle(const T *__restrict__ a, const Geom &g, const McLevels &lv, double iso,
                                            const uint8_t *s_tri, const int *s_e0, const int *s_e1, const int *s_ec, int32_t k,
                                            int32_t j, int32_t i, int idx, int rel, float *o) {
    const int32_t ka = k - (int32_t)g.pb, ja = ((int32_t)g.NY - 1 - j) - (int32_t)g.pxy; // source row of corner (dy=0, dz=0)
    const int32_t ia = i - (int32_t)g.pxy;
    const bool fast = ka >= 0 && ka + 1 < (int32_t)g.nz && ja - 1 >= 0 && ja < (int32_t)g.ny && ia >= 0 &&
                      ia + 1 < (int32_t)g.nx;
    double s0[3], s1[3];
    int ec[3];
#pragma unroll
    for (int v = 0; v < 3; v++) ec[v] = s_tri[idx * 15 + 3 * rel + v];
    double ttl[3];
    if (LEVELS) {
#pragma unroll
        for (int v = 0; v < 3; v++) {
            const int c = s_ec[ec[v]];
            const int ax = c & 3, bx = (c >> 2) & 1, by = (c >> 3) & 1, bz = (c >> 4) & 1;
            const int lo = bx + 2 * by + 4 * bz; // corner numbers of the edge's ends in the case index
            const bool in0 = (idx >> lo) & 1, in1 = (idx >> (lo + (1 << ax))) & 1; // exactly one of them is inside
            // the inside end is a source voxel (padding is never inside a from_binary piece): its bit of `sel`
            const int32_t kk = k + bz + (in1 && ax == 2), jj = j + by + (in1 && ax == 1), ii = i + bx + (in1 && ax == 0);
            const int64_t sk = kk - (int32_t)g.pb, sj = ((int32_t)g.NY - 1 - jj) - (int32_t)g.pxy, si = ii - (int32_t)g.pxy;
            const uint64_t wsel = lv.sel[(sk * g.ny + sj) * g.ws + (si >> 6)];
            ttl[v] = lv.tt[(in0 ? 2 : 0) + (int)((wsel >> (si & 63)) & 1ull)];
            ec[v] = c;
        }
    } else if (fast) { // interior cell: one base pointer, six byte/short gathers at table offsets, issued back to back
        const T *cell = a + ((int64_t)ka * g.ny + ja) * g.nx + ia;
#pragma unroll
        for (int v = 0; v < 3; v++) {
            s0[v] = (double)cell[s_e0[ec[v]]];
            s1[v] = (double)cell[s_e1[ec[v]]];
        }
#pragma unroll
        for (int v = 0; v < 3; v++) ec[v] = s_ec[ec[v]];
    } else {
#pragma unroll
        for (int v = 0; v < 3; v++) {
            const int c = s_ec[ec[v]];
            const int ax = c & 3, bx = (c >> 2) & 1, by = (c >> 3) & 1, bz = (c >> 4) & 1;
            s0[v] = mc_at(a, g, k + bz, j + by, i + bx);
            s1[v] = mc_at(a, g, k + bz + (ax == 2), j + by + (ax == 1), i + bx + (ax == 0));
            ec[v] = c;
        }
    }
#pragma unroll
    for (int v = 0; v < 3; v++) {
        const int ax = ec[v] & 3, bx = (ec[v] >> 2) & 1, by = (ec[v] >> 3) & 1, bz = (ec[v] >> 4) & 1;
        const double tt = LEVELS ? ttl[v] : (iso - s0[v]) / (s1[v] - s0[v]);
        double p0 = (double)(i + bx - (int32_t)g.pxy);
        double p1 = (double)(j + by - (int32_t)g.yoff);
        double p2 = (double)(k + bz + g.zoff);
        if (ax == 0) p0 += tt;
        else if (ax == 1) p1 += tt;
        else p2 += tt;
        o[3 * v + 0] = (float)(g.sx * p0);
        o[3 * v + 1] = (float)(g.sy * p1);
        o[3 * v + 2] = (float)(g.sz * p2);
    }
}

template <typename T, bool LEVELS>
__global__ __launch_bounds__(256) void k_mc_emit(const T *__restrict__ a, Geom g, double iso0, double iso1,
                                                 const uint64_t *__restrict__ split_dev,
                                                 const uint64_t *__restrict__ total_dev,
                                                 const uint64_t *__restrict__ list, uint64_t cap,
                                                 float *__restrict__ tris, McLevels lv) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tri[256 * 15];
    __shared__ __attribute__((aligned(16))) float s_out[256 * 9];
    const int tid = threadIdx.x;
    // The grid covers the output CAPACITY; how many triangles there really are (and where iso 1's begin) is read from
    // the scan's result on the device, so the host can queue this kernel before it knows the count.
    const uint64_t total = *total_dev, split = *split_dev;
    const uint64_t ntris = total < cap ? total : cap;
    const uint64_t T0 = (uint64_t)blockIdx.x * 256;
    if (T0 >= ntris) return;
    // the triangle table (3 840 bytes, rows of 15) as 960 dwords, layout unchanged
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q * 256 + tid < 960) ((uint32_t *)s_tri)[q * 256 + tid] = ((const uint32_t *)&MC_TRI[0][0])[q * 256 + tid];
    const uint64_t T_ = T0 + tid;
    const double iso = T_ < split ? iso0 : iso1;
    // per-edge constants, once per workgroup: voxel offsets of the edge's two end points relative to the cell's corner
    // (0,0,0) (dy runs against the source rows: the Y flip) and the end point / axis codes.  |offset| <= ny*nx + nx + 1.
    __shared__ int s_e0[12], s_e1[12], s_ec[12];
    const int plane32 = (int)(g.ny * g.nx), nx32 = (int)g.nx;
    if (tid < 12) {
        int ax, bx, by, bz;
        edge_decode(tid, ax, bx, by, bz);
        const int o0 = bz * plane32 - by * nx32 + bx;
        s_e0[tid] = o0;
        s_e1[tid] = o0 + (ax == 2 ? plane32 : (ax == 1 ? -nx32 : 1));
        s_ec[tid] = ax | (bx << 2) | (by << 3) | (bz << 4);
    }
    const bool live = T_ < ntris;
    const uint64_t d = live ? list[T_] : 0ull; // (a non-temporal load here measured the same)
    __syncthreads();
    if (live) {
        const int b = (int)(d >> 11) & 63, idx = (int)(d >> 3) & 255, rel = (int)d & 7;
        const uint32_t w = (uint32_t)(d >> 17) & 0x7fffu;
        const int32_t k = (int32_t)(d >> 48), j = (int32_t)((d >> 32) & 0xffffull);
        mc_triangle<T, LEVELS>(a, g, lv, iso, s_tri, s_e0, s_e1, s_ec, k, j, (int32_t)w * 64 + b, idx, rel, s_out + tid * 9);
    }
    __syncthreads();
    const uint32_t nt_chunk = ntris - T0 < 256 ? (uint32_t)(ntris - T0) : 256u;
    float *dst = tris + T0 * 9;
    if (nt_chunk == 256 && ((uintptr_t)tris & 15) == 0) { // a whole chunk (9 216 bytes, 16-byte aligned): 576 16-byte stores
        typedef float float4_t __attribute__((ext_vector_type(4)));
        float4_t *d4 = (float4_t *)dst;
        const float4_t *s4 = (const float4_t *)s_out;
        // non-temporal: the soup is written once and never read by the pipeline -- keeping it out of the L2 leaves the cache to the
        // list and the planes (k_mc_list + k_mc_emit 120 -> 105 us on the bench surface, step 0.440 -> 0.419 ms)
        __builtin_nontemporal_store(s4[tid], &d4[tid]);
        __builtin_nontemporal_store(s4[tid + 256], &d4[tid + 256]);
        if (tid < 64) __builtin_nontemporal_store(s4[tid + 512], &d4[tid + 512]);
    } else {
        for (uint32_t f = tid; f < nt_chunk * 9; f += 256) dst[f] = s_out[f];
    }
}

// ---- 5. the surface in ONE launch: count, offsets and emit without per-word counts, scan launch or triangle list ------------
// (VERDICT r4 item 5 / DESIGN section 9.4.)  The four-launch path walks the cell words twice (k_mc_count, k_mc_list), writes a
// 54 MB descriptor list and reads it straight back; its passes are bound by the life time of 9 259 short workgroups each.  Here
// a workgroup owns 256 cell words from the corner loads to the last store:
//   (a) corner words -> active-cell masks, active words packed to the first lanes (as k_mc_count does);
//   (b) those lanes walk their cells once for the word's triangle count -> block sum;
//   (c) the block's place in the output by a decoupled look-back over one status word per workgroup (flag | value, one
//       agent-scope store / load each: wave 0 inspects 64 predecessors per step; workgroups are dispatched in index order, so a
//       predecessor is always running or done) -- output order stays the oracle's (cell raster order);
//   (d) the lanes walk their cells again and write one 32-bit descriptor per triangle into an LDS window (slot | cell | case |
//       triangle-in-case; windows of MCF_WIN descriptors, one window for all but the densest blocks);
//   (e) one lane per triangle of the window, 256 at a time: the same arithmetic as k_mc_emit (mc_triangle), nine floats staged
//       in LDS at the output's own 16-byte phase, so the chunk leaves as aligned non-temporal 16-byte stores although a block's
//       first triangle starts anywhere (head / tail dwords go out one by one).
// The last workgroup leaves the total where k_mc_scan would (boff[nblocks]) for ivx_dev_mc_total.  One iso-value.
// MEASURED (round 5, profiles/r05/r05_mc_one_launch.md): same soup bit for bit, but 239 us against the 158 us of the four launches
// at 512^3 (1024^3: 1.29 vs 1.26 ms).  Decomposed by leaving parts out: phases (a)-(d) 74 us, the look-back 12 us, stores 21 us,
// and phase (e)'s arithmetic 130 us -- the per-chunk emit is a ~6 us latency chain (plane gathers, LDS, fp64) that the list-driven
// k_mc_emit hides behind 24 700 independent one-chunk workgroups, while here a workgroup's chunks queue up behind each other and
// behind its own barriers.  Kept as an opt-in (IVX_MC_ONE_LAUNCH=1; tests run both paths), not the default.
constexpr int MCF_WIN = 1024;
constexpr uint64_t MCF_VAL = (1ull << 40) - 1ull;
__device__ __forceinline__ void mcf_publish(unsigned long long *st, uint64_t flag, uint64_t v) {
    __hip_atomic_store(st, (unsigned long long)((flag << 40) | v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T, bool LEVELS>
__global__ __launch_bounds__(256) void k_mc_fused(const uint64_t *__restrict__ bits, const T *__restrict__ a, Geom g, size_t nwords,
                                                  uint64_t pbits, double iso, unsigned long long *state /* zeroed, one per workgroup */,
                                                  uint32_t *ticket /* zeroed */, uint64_t *__restrict__ total_out, uint64_t cap,
                                                  float *__restrict__ tris, McLevels lv) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tri[256 * 15];
    // 19 KB of LDS per workgroup = eight workgroups per CU (at 34 KB and four per CU the kernel took 269 us against the 158 us of
    // the four launches it replaces: a workgroup's phases are a chain of barriers and round trips that only other workgroups
    // hide).  The hand-over table of phase (a) -- per active word the four even corner words + the active mask -- is dead once
    // its lanes hold it in registers, so it shares its bytes with the staging buffer and the descriptor window of (d) / (e).
    constexpr int OUT_BYTES = (256 * 9 + 4) * 4;
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[OUT_BYTES + MCF_WIN * 4];
    static_assert(OUT_BYTES % 16 == 0 && OUT_BYTES + MCF_WIN * 4 >= 256 * 5 * 8, "LDS overlay");
    float *s_out = (float *)s_raw;
    uint32_t *s_desc = (uint32_t *)(s_raw + OUT_BYTES);
    uint64_t(*s_c)[5] = (uint64_t(*)[5])s_raw;
    __shared__ uint8_t s_ntri[256];
    __shared__ uint8_t s_hi[256];    // bit 63 of the four odd corner words (cell 63's far corners)
    __shared__ uint16_t s_k[256], s_j[256], s_w[256];
    __shared__ uint32_t s_wcnt[4], s_part[4];
    __shared__ unsigned long long s_excl;
    __shared__ int s_e0[12], s_e1[12], s_ec[12];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // The look-back below waits for workgroups with smaller ids: a workgroup's id is therefore the ORDER IN WHICH IT STARTED (a
    // ticket), not blockIdx.x -- whatever order the hardware dispatches in, everybody a workgroup waits for is already running
    // (the dispatch order is not promised).
    __shared__ uint32_t s_bid;
    if (tid == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;
    s_ntri[tid] = MC_NTRI[tid];
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q * 256 + tid < 960) ((uint32_t *)s_tri)[q * 256 + tid] = ((const uint32_t *)&MC_TRI[0][0])[q * 256 + tid];
    const int plane32 = (int)(g.ny * g.nx), nx32 = (int)g.nx;
    if (tid < 12) {
        int ax, bx, by, bz;
        edge_decode(tid, ax, bx, by, bz);
        const int o0 = bz * plane32 - by * nx32 + bx;
        s_e0[tid] = o0;
        s_e1[tid] = o0 + (ax == 2 ? plane32 : (ax == 1 ? -nx32 : 1));
        s_ec[tid] = ax | (bx << 2) | (by << 3) | (bz << 4);
    }
    // (a) corners, active words packed
    const size_t wid = (size_t)bid * 256 + tid;
    uint64_t act = 0;
    Corner8 r;
    uint32_t k = 0, j = 0, w = 0;
    if (wid < nwords) {
        const uint32_t row = (uint32_t)wid / (uint32_t)g.WC;
        w = (uint32_t)wid - row * (uint32_t)g.WC;
        k = row / (uint32_t)(g.NY - 1);
        j = row - k * (uint32_t)(g.NY - 1);
        r = load_corners(bits, g, (int64_t)k, (int64_t)j, (int64_t)w, pbits);
        act = r.active;
    }
    const unsigned long long am = __ballot(act != 0);
    if (lane == 0) s_wcnt[wv] = (uint32_t)__popcll(am);
    __syncthreads();
    uint32_t before = 0;
    for (int q = 0; q < wv; q++) before += s_wcnt[q];
    const uint32_t nact = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    if (act) {
        const uint32_t slot = before + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
        uint32_t hi = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            s_c[slot][q] = r.c[2 * q];
            hi |= (uint32_t)(r.c[2 * q + 1] >> 63) << q;
        }
        s_c[slot][4] = act;
        s_hi[slot] = (uint8_t)hi;
        s_k[slot] = (uint16_t)k;
        s_j[slot] = (uint16_t)j;
        s_w[slot] = (uint16_t)w;
    }
    __syncthreads();
    // (b) triangle count of my word (lanes < nact own one active word each)
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, mine = 0;
    uint32_t hi = 0, n = 0;
    const auto case_at = [&](int b) -> int {
        if (b != 63)
            return (int)((c0 >> b) & 3ull) | ((int)((c1 >> b) & 3ull) << 2) | ((int)((c2 >> b) & 3ull) << 4) | ((int)((c3 >> b) & 3ull) << 6);
        return (int)(c0 >> 63) | ((int)(hi & 1u) << 1) | ((int)(c1 >> 63) << 2) | ((int)(hi >> 1 & 1u) << 3) | ((int)(c2 >> 63) << 4) |
               ((int)(hi >> 2 & 1u) << 5) | ((int)(c3 >> 63) << 6) | ((int)(hi >> 3 & 1u) << 7);
    };
    if ((uint32_t)tid < nact) {
        c0 = s_c[tid][0]; c1 = s_c[tid][1]; c2 = s_c[tid][2]; c3 = s_c[tid][3];
        mine = s_c[tid][4];
        hi = s_hi[tid];
        uint64_t a2 = mine;
        while (a2) {
            const int b = __builtin_ctzll(a2);
            a2 &= a2 - 1;
            n += s_ntri[case_at(b)];
        }
    }
    uint32_t inc = n; // inclusive scan of the slots' counts inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_part[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int q = 0; q < wv; q++) wbase += s_part[q];
    const uint32_t S = s_part[0] + s_part[1] + s_part[2] + s_part[3]; // the block's triangles
    const uint32_t mypos = wbase + inc - n; // first triangle of my word, relative to the block
    // (c) where the block's triangles go: decoupled look-back, by the LAST wave (it rarely owns active words) while the other
    // waves already write the first window's descriptors, which are block-relative and need no offset
    if (wv == 3) {
        if (lane == 0) mcf_publish(&state[bid], bid == 0 ? 2ull : 1ull, (uint64_t)S);
        uint64_t excl = 0;
        if (bid > 0) {
            int64_t look = (int64_t)bid - 1;
            while (true) {
                const int64_t idx = look - lane;
                uint64_t v = 2ull << 40; // before block 0: an inclusive prefix of 0
                if (idx >= 0) {
                    v = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while ((v >> 40) == 0) { // (back off: two thousand resident workgroups poll the same few lines)
                        __builtin_amdgcn_s_sleep(16);
                        v = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                const unsigned long long pm = __ballot((v >> 40) == 2ull);
                const int stop = pm ? __builtin_ctzll(pm) : 64; // nearest predecessor whose inclusive prefix is known
                uint64_t contrib = lane <= stop ? (v & MCF_VAL) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
                excl += contrib;
                if (pm) break;
                look -= 64;
            }
            if (lane == 0) mcf_publish(&state[bid], 2ull, excl + S);
        }
        if (lane == 0) {
            s_excl = excl;
            if (bid == gridDim.x - 1) *total_out = excl + S;
        }
    }
    const bool vec_ok = ((uintptr_t)tris & 15) == 0;
    uint64_t E = 0;
    // (d) + (e), window by window
    for (uint32_t base = 0; base < S; base += MCF_WIN) {
        const uint32_t wend = base + MCF_WIN < S ? base + MCF_WIN : S;
        if ((uint32_t)tid < nact && mypos < wend && mypos + n > base) {
            uint64_t a2 = mine;
            uint32_t pos = mypos;
            while (a2 && pos < wend) {
                const int b = __builtin_ctzll(a2);
                a2 &= a2 - 1;
                const int idx = case_at(b);
                const uint32_t nt = s_ntri[idx];
                if (pos + nt > base) {
                    const uint32_t d0 = ((uint32_t)tid << 17) | ((uint32_t)b << 11) | ((uint32_t)idx << 3);
#pragma unroll
                    for (uint32_t t = 0; t < MC_MAX_TRI; t++)
                        if (t < nt && pos + t >= base && pos + t < wend) s_desc[pos + t - base] = d0 | t;
                }
                pos += nt;
            }
        }
        __syncthreads();
        if (base == 0) {
            E = s_excl;
            if (E >= cap) return; // (uniform) nothing of this block fits the caller's buffer; the total still reports the need
        }
        for (uint32_t cb = base; cb < wend; cb += 256) {
            const uint64_t G = E + cb; // global index of the chunk's first triangle
            if (G >= cap) break;       // (uniform)
            uint32_t ntc = wend - cb < 256 ? wend - cb : 256u;
            if (G + ntc > cap) ntc = (uint32_t)(cap - G);
            const uint32_t ph = (uint32_t)((G * 9ull) & 3ull); // the chunk's first float inside its 16-byte line
            if ((uint32_t)tid < ntc) {
                const uint32_t d = s_desc[cb - base + tid];
                const uint32_t slot = d >> 17;
                const int b = (int)(d >> 11) & 63, idx = (int)(d >> 3) & 255, rel = (int)d & 7;
                mc_triangle<T, LEVELS>(a, g, lv, iso, s_tri, s_e0, s_e1, s_ec, (int32_t)s_k[slot], (int32_t)s_j[slot],
                                       (int32_t)s_w[slot] * 64 + b, idx, rel, s_out + ph + tid * 9);
            }
            __syncthreads();
            const uint32_t nf = ntc * 9;
            float *dst = tris + G * 9ull - ph; // 16-byte aligned when tris is
            if (vec_ok) {
                typedef float float4_t __attribute__((ext_vector_type(4)));
                const uint32_t nvec = (ph + nf + 3) / 4;
                for (uint32_t v = tid; v < nvec; v += 256) {
                    const uint32_t f0 = v * 4;
                    if (f0 >= ph && f0 + 4 <= ph + nf) {
                        __builtin_nontemporal_store(((const float4_t *)s_out)[v], &((float4_t *)dst)[v]);
                    } else {
#pragma unroll
                        for (uint32_t q = 0; q < 4; q++)
                            if (f0 + q >= ph && f0 + q < ph + nf) dst[f0 + q] = s_out[f0 + q];
                    }
                }
            } else {
                for (uint32_t f = tid; f < nf; f += 256) dst[ph + f] = s_out[ph + f];
            }
            __syncthreads();
        }
    }
}

static std::map<const void *, uint64_t> g_split; // scratch -> number of iso-0 triangles (two-iso pieces)
static std::map<const void *, uint32_t> g_vsplit; // scratch -> number of iso-0 vertices (indexed mesh)
static std::mutex g_split_mu;
// scratch -> inside plane handed in by ivx_dev_mc_count_bits (read in place by the later list / indexed passes of the
// same piece instead of being copied into the scratch); an ivx_dev_mc_count on the same scratch forgets it
static std::map<const void *, const uint64_t *> g_ext_bits;
// ivx_dev_mc_list: scratch -> (list buffer, capacity) of a triangle list built ahead of the emit, valid until the next
// count on that scratch; list buffer (a per-stream workspace shared by every piece on that stream) -> the scratch whose
// descriptors it currently holds
struct ListBuilt {
    const void *list;
    int64_t max_tris;
};
static std::map<const void *, ListBuilt> g_list_built;
static std::map<const void *, const void *> g_list_owner;
// may the list pass be skipped: was this very buffer filled for `scratch`, with room for all the caller will read?
static bool list_ready(const void *scratch, const void *d_list, int64_t max_tris) {
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_list_built.find(scratch);
    auto ow = g_list_owner.find(d_list);
    const bool ok = it != g_list_built.end() && it->second.list == d_list && max_tris <= it->second.max_tris &&
                    ow != g_list_owner.end() && ow->second == scratch;
    if (!ok) g_list_owner[d_list] = nullptr; // the caller is about to overwrite it
    return ok;
}

static const uint64_t *mc_bits_ptr(const void *scratch, const Scratch &s, int q) {
    if (q == 0) {
        std::lock_guard<std::mutex> lk(g_split_mu);
        auto it = g_ext_bits.find(scratch);
        if (it != g_ext_bits.end()) return it->second;
    }
    return (const uint64_t *)((const char *)scratch + s.off_bits) + (size_t)q * s.bits_words;
}

static inline uint64_t pad_bits(const ivx_mc_params *p, int q) { return p->pad_value >= p->iso[q] ? ~0ull : 0ull; }

template <typename T>
static int run_bits(const ivx_mc_params *p, const Geom &g, const Scratch &s, const void *a, uint8_t *b0, double iso0,
                    double iso1, hipStream_t st) {
    constexpr int V = 16 / sizeof(T);
    const int64_t nrows_src = g.nz * g.ny;
    const int64_t total = nrows_src * (g.ws * 64 / V);
    if (total == 0) return IVX_OK;
    const int64_t blocks = ivx::cdiv(total, 256 * 4);
    const int grid = (int)(blocks < 1 ? 1 : (blocks < 16384 ? blocks : 16384));
    uint8_t *b1 = b0 + s.bits_words * 8;
    if (p->niso == 2)
        hipLaunchKernelGGL((k_mc_bits<T, 2>), dim3(grid), dim3(256), 0, st, (const T *)a, nrows_src, g.nx, g.ws,
                           iso0, iso1, b0, b1);
    else
        hipLaunchKernelGGL((k_mc_bits<T, 1>), dim3(grid), dim3(256), 0, st, (const T *)a, nrows_src, g.nx, g.ws,
                           iso0, 0.0, b0, b0);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

static int run_list(const ivx_mc_params *p, const Geom &g, const Scratch &s, const char *scratch, void *d_list,
                    int64_t max_tris, hipStream_t st) {
    const uint64_t *boff = (const uint64_t *)(scratch + s.off_boff);
    for (int q = 0; q < p->niso; q++) {
        const uint64_t *bits = mc_bits_ptr(scratch, s, q);
        const uint16_t *counts = (const uint16_t *)(scratch + s.off_counts) + (size_t)q * s.nwords;
        hipLaunchKernelGGL(k_mc_list, dim3((unsigned)s.nblocks), dim3(256), 0, st, bits, g, s.nwords, pad_bits(p, q), counts,
                           boff + (size_t)q * s.nblocks, (uint64_t *)d_list, (uint64_t)max_tris);
        IVX_LAUNCH_CHECK();
    }
    return IVX_OK;
}

template <typename T>
static int run_emit(const ivx_mc_params *p, const Geom &g, const Scratch &s, const void *a, const char *scratch,
                    float *tris, int64_t max_tris, hipStream_t st, const McLevels *lv = nullptr) {
    if (s.nblocks == 0) return IVX_OK;
    IVX_REQUIRE(s.nwords < 0xffffffffull, IVX_EINVAL, "mc: piece too large for 32-bit cell-word ids");
    const uint64_t *boff = (const uint64_t *)(scratch + s.off_boff);
    void *d_list;
    int rc = ivx::ws_get_s(ivx::WS_MCLIST, st, (size_t)max_tris * 8 + 64, &d_list);
    if (rc) return rc;
    // iso 1's triangles follow iso 0's: boff is one scan over [iso0 blocks | iso1 blocks], so both list passes write
    // disjoint ranges of ONE list and one flat emit covers all of it; the kernel reads the total (boff[nb]) and the
    // iso-0 / iso-1 split (boff[nblocks]) on the device
    const size_t nb = s.nblocks * (size_t)p->niso;
    if (!list_ready(scratch, d_list, max_tris)) // not built ahead by ivx_dev_mc_list
        if ((rc = run_list(p, g, s, scratch, d_list, max_tris, st))) return rc;
    if (lv)
        hipLaunchKernelGGL((k_mc_emit<T, true>), dim3((unsigned)ivx::cdiv(max_tris, (int64_t)256)), dim3(256), 0, st, (const T *)a, g,
                           p->iso[0], p->iso[1], boff + (p->niso == 2 ? s.nblocks : nb), boff + nb, (const uint64_t *)d_list,
                           (uint64_t)max_tris, tris, *lv);
    else
        hipLaunchKernelGGL((k_mc_emit<T, false>), dim3((unsigned)ivx::cdiv(max_tris, (int64_t)256)), dim3(256), 0, st, (const T *)a, g,
                           p->iso[0], p->iso[1], boff + (p->niso == 2 ? s.nblocks : nb), boff + nb, (const uint64_t *)d_list,
                           (uint64_t)max_tris, tris, McLevels{nullptr, 0.0, 0.0, 0.0, {0.0, 0.0, 0.0, 0.0}});
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// =====================================================================================================================
// Indexed mesh ("point merge" of join_process_surface, invesalius/data/surface_process.py:229-268: the reference appends
// the pieces and runs vtkCleanPolyData to merge coincident points).  Here the merge needs no hashing or sorting: a
// vertex IS a grid edge whose end points differ in the inside-bit plane, so
//   crossing planes   cx = P ^ (P >> 1 | carry), cy = P(j) ^ P(j+1), cz = P(k) ^ P(k+1)   (P = padded point words)
//   vertex id         = (scan of popcounts over point words in raster order) + rank of the edge inside its word
//                       (x edges first, then y, then z)
//   k_mci_vertices    one interpolation per UNIQUE vertex (3.2 M instead of 19 M for the bench surface)
//   k_mci_faces       one lane per triangle of the flat list: three edge -> id look-ups (bit planes + popcounts)
// verts[faces] reproduces the soup of ivx_dev_mc_emit bit for bit.
// =====================================================================================================================
struct Cross {
    uint64_t cx, cy, cz; // regular crossings: bit b = the edge leaving point (64w+b, jf, k) in +x / +y / +z
    uint64_t cp;         // point vertices (only with POINTS): the point's value IS the iso-value and a neighbour is outside
    uint64_t e0, ex, ey, ez; // "value == iso" at the point itself and at its +x / +y / +z neighbour
};
// S = inside plane (value >= iso), Q = strictly-inside plane (value > iso); E = S & ~Q marks points sitting exactly
// on the iso-value.  A crossing edge with such an end point puts its vertex ON that grid point (t is exactly 0 or 1),
// and every such edge around the point yields the same position: those become ONE "point vertex", owned by the point.
template <bool POINTS>
__device__ __forceinline__ Cross crossings(const uint64_t *__restrict__ S, const uint64_t *__restrict__ Q, const Geom &g,
                                           int64_t k, int64_t jf, int64_t w, uint64_t pbits, uint64_t qbits) {
    uint64_t p0, p0n, py, pyn, pz, pzn, q0, q0n, qy, qyn, qz, qzn;
    padded_pair(S, g, k, jf, w, pbits, p0, p0n);
    padded_pair(S, g, k, jf + 1, w, pbits, py, pyn);
    padded_pair(S, g, k + 1, jf, w, pbits, pz, pzn);
    padded_pair(Q, g, k, jf, w, qbits, q0, q0n);
    padded_pair(Q, g, k, jf + 1, w, qbits, qy, qyn);
    padded_pair(Q, g, k + 1, jf, w, qbits, qz, qzn);
    const int64_t rem = g.NX - w * 64;                 // points of this word
    const uint64_t pts = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
    const int64_t remx = g.NX - 1 - w * 64;            // x edges of this word (last point has none)
    const uint64_t xed = remx >= 64 ? ~0ull : (remx <= 0 ? 0ull : ((1ull << remx) - 1ull));
    const bool hasy = jf + 1 < g.NY, hasz = k + 1 < g.NZ;
    Cross c;
    c.e0 = p0 & ~q0 & pts;
    c.ex = (((p0 & ~q0) >> 1) | ((p0n & ~q0n) << 63)) & xed;
    c.ey = hasy ? (py & ~qy & pts) : 0ull;
    c.ez = hasz ? (pz & ~qz & pts) : 0ull;
    const uint64_t rx = (p0 ^ ((p0 >> 1) | (p0n << 63))) & xed;
    const uint64_t ry = hasy ? ((p0 ^ py) & pts) : 0ull;
    const uint64_t rz = hasz ? ((p0 ^ pz) & pts) : 0ull;
    c.cx = rx & ~(c.e0 | c.ex);
    c.cy = ry & ~(c.e0 | c.ey);
    c.cz = rz & ~(c.e0 | c.ez);
    c.cp = 0ull;
    if (POINTS && c.e0) {
        uint64_t prev = 0ull, t0, t1, rxm, rym = 0ull, rzm = 0ull;
        if (w > 0) {
            padded_pair(S, g, k, jf, w - 1, pbits, t0, t1);
            prev = t0 >> 63;
        }
        rxm = (p0 ^ ((p0 << 1) | prev)) & (w > 0 ? ~0ull : ~1ull);
        if (jf > 0) {
            padded_pair(S, g, k, jf - 1, w, pbits, t0, t1);
            rym = p0 ^ t0;
        }
        if (k > 0) {
            padded_pair(S, g, k - 1, jf, w, pbits, t0, t1);
            rzm = p0 ^ t0;
        }
        c.cp = c.e0 & (rx | ry | rz | rxm | rym | rzm);
    }
    return c;
}

// The crossing words of every point word are derived ONCE (k_mci_count) and kept as 32-byte records: the vertex pass reads
// its word's record, and a face corner is two gathers (record + vertex base) instead of six padded row pairs of the two
// planes (a dozen loads and the masking around them) per corner -- three corners per triangle, 394 M triangles at 2048^3.
struct __attribute__((aligned(32))) CrossRec {
    uint64_t cx, cy, cz, cp;
};
__global__ __launch_bounds__(256) void k_mci_count(const uint64_t *__restrict__ bits, const uint64_t *__restrict__ qb,
                                                   Geom g, int64_t npw, uint64_t pbits, uint64_t qbits,
                                                   uint32_t *__restrict__ vcnt, CrossRec *__restrict__ rec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npw; i += stride) {
        const int64_t w = i % g.WX, r = i / g.WX, jf = r % g.NY, k = r / g.NY;
        const Cross c = crossings<true>(bits, qb, g, k, jf, w, pbits, qbits);
        vcnt[i] = (uint32_t)(__popcll(c.cx) + __popcll(c.cy) + __popcll(c.cz) + __popcll(c.cp));
        rec[i] = CrossRec{c.cx, c.cy, c.cz, c.cp};
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_mci_vertices(const T *__restrict__ a, const CrossRec *__restrict__ rec, Geom g,
                                                      int64_t npw, double iso, const uint32_t *__restrict__ vbase,
                                                      uint32_t id0, float *__restrict__ verts, uint64_t max_verts) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t pw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pw < npw; pw += stride) {
        const CrossRec c = rec[pw];
        if (!(c.cx | c.cy | c.cz | c.cp)) continue;
        const int64_t w = pw % g.WX, r = pw / g.WX, jf = r % g.NY, k = r / g.NY;
        uint64_t id = (uint64_t)id0 + vbase[pw];
#pragma unroll
        for (int ax = 0; ax < 4; ax++) {
            uint64_t m = ax == 0 ? c.cx : (ax == 1 ? c.cy : (ax == 2 ? c.cz : c.cp));
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int64_t i = w * 64 + b;
                double p0 = (double)(i - g.pxy), p1 = (double)(jf - g.yoff), p2 = (double)(k + g.zoff);
                if (ax < 3) {
                    const double s0 = mc_at(a, g, k, jf, i);
                    const double s1 = mc_at(a, g, k + (ax == 2), jf + (ax == 1), i + (ax == 0));
                    const double tt = (iso - s0) / (s1 - s0);
                    if (ax == 0) p0 += tt;
                    else if (ax == 1) p1 += tt;
                    else p2 += tt;
                }
                if (id < max_verts) {
                    float *o = verts + id * 3;
                    o[0] = (float)(g.sx * p0);
                    o[1] = (float)(g.sy * p1);
                    o[2] = (float)(g.sz * p2);
                }
                id++;
            }
        }
    }
}

// k_mci_vertices for a uint8 mask whose bytes are KNOWN (McLevels: v_out outside the inside plane -- the padding too --,
// v_sel where `sel` has a bit, v_in elsewhere inside): no voxel is read.  Which end of a crossing edge is inside is the
// point's bit of the padded inside row, the interpolation factor one of four constants, and no point sits on the iso-value
// (no point vertices: cp is empty).  Same vertices, same order, same bits as k_mci_vertices on that mask.
__global__ __launch_bounds__(256) void k_mci_vertices_levels(const uint64_t *__restrict__ bits, const CrossRec *__restrict__ rec,
                                                             Geom g, int64_t npw, uint64_t pbits, McLevels lv,
                                                             const uint32_t *__restrict__ vbase, uint32_t id0,
                                                             float *__restrict__ verts, uint64_t max_verts) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t pw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pw < npw; pw += stride) {
        const CrossRec c = rec[pw];
        if (!(c.cx | c.cy | c.cz)) continue;
        const int64_t w = pw % g.WX, r = pw / g.WX, jf = r % g.NY, k = r / g.NY;
        uint64_t p0, p0n;
        padded_pair(bits, g, k, jf, w, pbits, p0, p0n);
        uint64_t id = (uint64_t)id0 + vbase[pw];
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            uint64_t m = ax == 0 ? c.cx : (ax == 1 ? c.cy : c.cz);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int64_t i = w * 64 + b;
                const bool in0 = (p0 >> b) & 1ull;
                int sel = 0;
                if (lv.sel) { // the inside end is a source voxel: its bit of the selection plane
                    const int64_t ii = i + (!in0 && ax == 0), jj = jf + (!in0 && ax == 1), kk = k + (!in0 && ax == 2);
                    const int64_t sk = kk - g.pb, sj = (g.NY - 1 - jj) - g.pxy, si = ii - g.pxy;
                    sel = (int)((lv.sel[(sk * g.ny + sj) * g.ws + (si >> 6)] >> (si & 63)) & 1ull);
                }
                const double tt = lv.tt[(in0 ? 2 : 0) + sel];
                double q0 = (double)(i - g.pxy), q1 = (double)(jf - g.yoff), q2 = (double)(k + g.zoff);
                if (ax == 0) q0 += tt;
                else if (ax == 1) q1 += tt;
                else q2 += tt;
                if (id < max_verts) {
                    float *o = verts + id * 3;
                    o[0] = (float)(g.sx * q0);
                    o[1] = (float)(g.sy * q1);
                    o[2] = (float)(g.sz * q2);
                }
                id++;
            }
        }
    }
}

// id of the vertex on the edge leaving point (i, jf, k) along axis ax -- a crossing edge of a triangle, so exactly one of
// three holds: the edge's bit is set in its point word's c{x,y,z} (a regular crossing: rank among the word's crossings);
// or the point's bit is set in cp (the point's value IS the iso-value: the vertex is that point's, cp = e0 & "some crossing
// leaves or reaches the point", and this very edge is such a crossing); or the far end point is the one on the iso-value.
__device__ __forceinline__ uint32_t vertex_id(const CrossRec *__restrict__ rec, const Geom &g, const uint32_t *__restrict__ vbase,
                                              int64_t k, int64_t jf, int64_t i, int ax) {
    int64_t w = i >> 6;
    int b = (int)(i & 63);
    int64_t pw = (k * g.NY + jf) * g.WX + w;
    CrossRec c = rec[pw];
    const uint64_t below = (1ull << b) - 1ull;
    if (((ax == 0 ? c.cx : (ax == 1 ? c.cy : c.cz)) >> b) & 1ull) {
        uint32_t rank;
        if (ax == 0) rank = (uint32_t)__popcll(c.cx & below);
        else if (ax == 1) rank = (uint32_t)(__popcll(c.cx) + __popcll(c.cy & below));
        else rank = (uint32_t)(__popcll(c.cx) + __popcll(c.cy) + __popcll(c.cz & below));
        return vbase[pw] + rank;
    }
    // the vertex sits on a grid point: it is that point's vertex
    if (!((c.cp >> b) & 1ull)) {
        if (ax == 0) i++;
        else if (ax == 1) jf++;
        else k++;
        w = i >> 6;
        b = (int)(i & 63);
        pw = (k * g.NY + jf) * g.WX + w;
        c = rec[pw];
    }
    const uint32_t rank = (uint32_t)(__popcll(c.cx) + __popcll(c.cy) + __popcll(c.cz) + __popcll(c.cp & ((1ull << b) - 1ull)));
    return vbase[pw] + rank;
}

__global__ __launch_bounds__(256) void k_mci_faces(const CrossRec *__restrict__ rec, Geom g,
                                                   const uint32_t *__restrict__ vbase, uint32_t id0,
                                                   const uint64_t *__restrict__ list, uint64_t ntris,
                                                   int32_t *__restrict__ faces) {
    __shared__ uint8_t s_tri[256 * 16];
#pragma unroll
    for (int q = 0; q < 15; q++) s_tri[threadIdx.x * 16 + q] = MC_TRI[threadIdx.x][q];
    __syncthreads();
    const uint64_t T_ = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (T_ >= ntris) return;
    const uint64_t d = list[T_];
    const int b = (int)(d >> 11) & 63, idx = (int)(d >> 3) & 255, rel = (int)d & 7;
    const uint32_t w = (uint32_t)(d >> 17) & 0x7fffu;
    const int64_t k = (int64_t)(d >> 48), j = (int64_t)((d >> 32) & 0xffffull);
    const int64_t i = (int64_t)w * 64 + b;
#pragma unroll
    for (int v = 0; v < 3; v++) {
        const int e = s_tri[idx * 16 + 3 * rel + v];
        int ax, bx, by, bz;
        edge_decode(e, ax, bx, by, bz);
        faces[T_ * 3 + v] = (int32_t)(id0 + vertex_id(rec, g, vbase, k + bz, j + by, i + bx, ax));
    }
}

// =====================================================================================================================
// Cross-slab stitch on the device (the vtkAppendPolyData + vtkCleanPolyData of join_process_surface,
// invesalius/data/surface_process.py:229-268, across the Z-slabs of SURVEY.md 8e).  Rank r's TOP point plane and rank r+1's
// BOTTOM point plane are the same slice of voxels, so both carry the vertices of that plane's x / y edges (and its point
// vertices).  A vertex IS a grid edge: the two copies are matched by edge identity -- same point word, same kind, same bit
// -- never by comparing floats:
//   k_mci_sig      per point word of a plane: (first local vertex id, cx, cy, cp)                 32 bytes per word
//   k_mci_match    bottom plane of this piece AND the signature received from below -> the copies to drop, per word + total
//   k_mci_gid0     global id of every vertex of the bottom plane's words: a dropped copy takes the id its twin has in the
//                  rank below, a kept one moves down by the copies dropped before it
//   k_mci_stitch_faces / _verts   faces -> global ids, vertices compacted; everything above the bottom plane just shifts
// Global numbering: rank r's kept vertices follow rank r-1's, base_r = sum over q < r of (V_q - D_q); the (V, D) pairs
// travel through ONE all-gather of 8 bytes per rank.  The result equals the host stitch (tests/_stitch_ref.py) array for
// array.
// =====================================================================================================================
struct PlaneSig {
    uint32_t vbase, pad;
    uint64_t cx, cy, cp;
};
static_assert(sizeof(PlaneSig) == 32, "plane signatures travel as raw bytes");

__global__ __launch_bounds__(256) void k_mci_sig(const uint64_t *__restrict__ bits, const uint64_t *__restrict__ qb, Geom g,
                                                 uint64_t pbits, uint64_t qbits, const uint32_t *__restrict__ vbase, int64_t k,
                                                 PlaneSig *__restrict__ sig) {
    const int64_t nwp = g.NY * g.WX;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwp) return;
    const int64_t w = i % g.WX, jf = i / g.WX;
    const Cross c = crossings<true>(bits, qb, g, k, jf, w, pbits, qbits);
    PlaneSig sgn;
    sgn.vbase = vbase[k * nwp + i];
    sgn.pad = 0;
    sgn.cx = c.cx;
    sgn.cy = c.cy;
    sgn.cp = c.cp;
    sig[i] = sgn;
}

// vd[0] = this piece's vertex count, vd[1] (zeroed before) += copies dropped; rmcnt[word] = copies dropped in that word
__global__ __launch_bounds__(256) void k_mci_match(const uint64_t *__restrict__ bits, const uint64_t *__restrict__ qb, Geom g,
                                                   uint64_t pbits, uint64_t qbits, const PlaneSig *__restrict__ nbr,
                                                   uint32_t *__restrict__ rmcnt, uint32_t *vd, uint32_t nverts) {
    const int64_t nwp = g.NY * g.WX;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) vd[0] = nverts;
    uint32_t n = 0;
    if (i < nwp && nbr) {
        const int64_t w = i % g.WX, jf = i / g.WX;
        const Cross c = crossings<true>(bits, qb, g, 0, jf, w, pbits, qbits);
        const PlaneSig o = nbr[i];
        n = (uint32_t)(__popcll(c.cx & o.cx) + __popcll(c.cy & o.cy) + __popcll(c.cp & o.cp));
    }
    if (i < nwp) rmcnt[i] = n;
    uint32_t sum = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&vd[1], sum);
}

__device__ __forceinline__ void stitch_bases(const uint32_t *__restrict__ vd_all, int rank, uint32_t &base, uint32_t &base_below,
                                             uint32_t &d_below) {
    uint32_t b = 0;
    base_below = 0;
    d_below = 0;
    for (int q = 0; q < rank; q++) {
        if (q == rank - 1) {
            base_below = b;
            d_below = vd_all[2 * q + 1];
        }
        b += vd_all[2 * q] - vd_all[2 * q + 1];
    }
    base = b;
}

// rmoff = exclusive scan of rmcnt over the bottom plane's words
__global__ __launch_bounds__(256) void k_mci_gid0(const uint64_t *__restrict__ bits, const uint64_t *__restrict__ qb, Geom g,
                                                  uint64_t pbits, uint64_t qbits, const uint32_t *__restrict__ vbase,
                                                  const PlaneSig *__restrict__ nbr, const uint32_t *__restrict__ rmoff,
                                                  const uint32_t *__restrict__ vd_all, int rank, uint32_t *__restrict__ gid0) {
    const int64_t nwp = g.NY * g.WX;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwp) return;
    const int64_t w = i % g.WX, jf = i / g.WX;
    const Cross c = crossings<true>(bits, qb, g, 0, jf, w, pbits, qbits);
    if (!(c.cx | c.cy | c.cz | c.cp)) return;
    uint32_t base, base_below, d_below;
    stitch_bases(vd_all, rank, base, base_below, d_below);
    PlaneSig o;
    o.vbase = 0; o.pad = 0; o.cx = 0; o.cy = 0; o.cp = 0;
    if (nbr) o = nbr[i];
    uint32_t local = vbase[i];
    uint32_t kept = local - rmoff[i]; // position among this piece's kept vertices
    // the rank below numbers the word's vertices cx, cy, (no cz on its top plane), cp
    const uint32_t ocx = (uint32_t)__popcll(o.cx), ocy = (uint32_t)__popcll(o.cy);
#pragma unroll
    for (int ax = 0; ax < 4; ax++) {
        uint64_t m = ax == 0 ? c.cx : (ax == 1 ? c.cy : (ax == 2 ? c.cz : c.cp));
        const uint64_t om = ax == 0 ? o.cx : (ax == 1 ? o.cy : (ax == 2 ? 0ull : o.cp));
        const uint32_t obefore = ax == 0 ? 0u : (ax == 1 ? ocx : ocx + ocy);
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            if (om >> b & 1ull) // a copy: the twin's id in the rank below, after ITS dropped copies (all of which precede its top plane)
                gid0[local] = base_below + (o.vbase + obefore + (uint32_t)__popcll(om & ((1ull << b) - 1ull))) - d_below;
            else
                gid0[local] = base + kept++;
            local++;
        }
    }
}

__global__ __launch_bounds__(256) void k_mci_stitch_faces(int32_t *__restrict__ faces, int64_t n3, uint32_t p0,
                                                          const uint32_t *__restrict__ gid0, const uint32_t *__restrict__ vd_all,
                                                          int rank) {
    uint32_t base, bb, db;
    stitch_bases(vd_all, rank, base, bb, db);
    const uint32_t d = vd_all[2 * rank + 1];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += stride) {
        const uint32_t v = (uint32_t)faces[i];
        faces[i] = (int32_t)(v < p0 ? gid0[v] : base + v - d);
    }
}

__global__ __launch_bounds__(256) void k_mci_stitch_verts(const float *__restrict__ verts, int64_t nverts, uint32_t p0,
                                                          const uint32_t *__restrict__ gid0, const uint32_t *__restrict__ vd_all,
                                                          int rank, float *__restrict__ out) {
    uint32_t base, bb, db;
    stitch_bases(vd_all, rank, base, bb, db);
    const uint32_t d = vd_all[2 * rank + 1];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nverts; v += stride) {
        const uint32_t gid = v < (int64_t)p0 ? gid0[v] : base + (uint32_t)v - d;
        if (gid < base) continue; // a dropped copy: its twin lives in the rank below
        const uint32_t o = gid - base;
        out[3 * (size_t)o] = verts[3 * v];
        out[3 * (size_t)o + 1] = verts[3 * v + 1];
        out[3 * (size_t)o + 2] = verts[3 * v + 2];
    }
}

// per-stream workspace WS_MCV: strict[niso][bits_words] u64 | per iso: vbase[npw] u32, bsum[nsb], total[16] | per iso:
// crossing records[npw] (32 B each)
struct MciLayout {
    int64_t npw, nsb;
    size_t off_v, per_iso, off_rec, per_iso_rec, total;
};
static MciLayout mci_layout(const Geom &g, const Scratch &s, int niso) {
    MciLayout m;
    m.npw = g.NZ * g.NY * g.WX;
    m.nsb = ivx::cdiv(m.npw, 256 * 16);
    m.off_v = al256((size_t)niso * s.bits_words * 8 + 16);
    m.per_iso = al256(((size_t)m.npw + (size_t)m.nsb + 16) * 4);
    m.off_rec = m.off_v + (size_t)niso * m.per_iso;
    m.per_iso_rec = al256((size_t)m.npw * sizeof(CrossRec));
    m.total = m.off_rec + (size_t)niso * m.per_iso_rec;
    return m;
}
static inline uint64_t pad_qbits(const ivx_mc_params *p, int q) { return p->pad_value > p->iso[q] ? ~0ull : 0ull; }

} // namespace

extern "C" int ivx_dev_mc_scratch_bytes(const ivx_mc_params *p, size_t *nbytes) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    *nbytes = make_scratch(g, p->niso).total;
    return IVX_OK;
}

// classify + count + scan over the inside planes already sitting in scratch (queued, nothing comes back to the host)
static int mc_queue_count(const ivx_mc_params *p, const Geom &g, const Scratch &s, void *scratch_, hipStream_t st) {
    char *scratch = (char *)scratch_;
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        g_list_built.erase(scratch_); // new counts: a list built from the old ones is void
    }
    uint32_t *bsum = (uint32_t *)(scratch + s.off_bsum);
    uint64_t *boff = (uint64_t *)(scratch + s.off_boff);
    for (int q = 0; q < p->niso; q++) {
        const uint64_t *bits = mc_bits_ptr(scratch, s, q);
        uint16_t *counts = (uint16_t *)(scratch + s.off_counts) + (size_t)q * s.nwords;
        hipLaunchKernelGGL(k_mc_count, dim3((unsigned)s.nblocks), dim3(256), 0, st, bits, g, s.nwords, pad_bits(p, q),
                           counts, bsum + (size_t)q * s.nblocks);
        IVX_LAUNCH_CHECK();
    }
    const size_t nb = s.nblocks * (size_t)p->niso;
    hipLaunchKernelGGL(k_mc_scan, dim3((unsigned)std::max<size_t>(1, (nb + 16383) / 16384)), dim3(1024), 0, st, bsum, nb, boff);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
// the scan's total (and the iso-0 / iso-1 split) through the mailbox
static int mc_read_total(const ivx_mc_params *p, const Scratch &s, void *scratch_, int64_t *ntris, hipStream_t st) {
    char *scratch = (char *)scratch_;
    int rc;
    uint64_t *boff = (uint64_t *)(scratch + s.off_boff);
    const size_t nb = s.nblocks * (size_t)p->niso;
    uint32_t seq, tw[2];
    if (p->niso == 2) { // also fetch where iso 0's triangles end (= boff[nblocks]): the indexed-mesh path needs it
        uint32_t seq0, t0[2];
        if ((rc = ivx::mailbox_publish(boff + s.nblocks, 2, st, &seq0))) return rc;
        if ((rc = ivx::mailbox_wait(seq0, st, t0, 2))) return rc;
        std::lock_guard<std::mutex> lk(g_split_mu);
        g_split[scratch_] = ((uint64_t)t0[1] << 32) | t0[0];
    }
    if ((rc = ivx::mailbox_publish(boff + nb, 2, st, &seq))) return rc;
    if ((rc = ivx::mailbox_wait(seq, st, tw, 2))) return rc;
    *ntris = (int64_t)(((uint64_t)tw[1] << 32) | tw[0]);
    return IVX_OK;
}

static int mc_count_impl(const ivx_mc_params *p, const void *a, void *scratch_, int64_t *ntris, void *stream,
                         bool ntris_wanted) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    const Scratch s = make_scratch(g, p->niso);
    char *scratch = (char *)scratch_;
    hipStream_t st = ivx::S(stream);
    *ntris = 0;
    if (s.nwords == 0) return IVX_OK;
    IVX_REQUIRE(s.nblocks * (size_t)p->niso < 0x7fffffffull, IVX_EINVAL, "mc: piece too large for one launch");
    IVX_REQUIRE(s.nwords < 0xffffffffull, IVX_EINVAL, "mc: piece too large for 32-bit cell-word ids");
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        g_ext_bits.erase(scratch_); // this piece's planes are derived into the scratch
    }
    switch (p->dtype) {
    case IVX_U8: rc = run_bits<uint8_t>(p, g, s, a, (uint8_t *)(scratch + s.off_bits), p->iso[0], p->iso[1], st); break;
    case IVX_I16: rc = run_bits<int16_t>(p, g, s, a, (uint8_t *)(scratch + s.off_bits), p->iso[0], p->iso[1], st); break;
    default: rc = run_bits<uint16_t>(p, g, s, a, (uint8_t *)(scratch + s.off_bits), p->iso[0], p->iso[1], st); break;
    }
    if (rc) return rc;
    if ((rc = mc_queue_count(p, g, s, scratch_, st))) return rc;
    return ntris_wanted ? mc_read_total(p, s, scratch_, ntris, st) : IVX_OK;
}

extern "C" int ivx_dev_mc_count(const ivx_mc_params *p, const void *a, void *scratch_, int64_t *ntris, void *stream) {
    return mc_count_impl(p, a, scratch_, ntris, stream, true);
}
// queue the counting passes only; ivx_dev_mc_emit may follow at once with the CAPACITY of `tris` as max_tris (it reads
// the real count on the device); ivx_dev_mc_total fetches the count afterwards
extern "C" int ivx_dev_mc_count_async(const ivx_mc_params *p, const void *a, void *scratch_, void *stream) {
    int64_t unused = 0;
    return mc_count_impl(p, a, scratch_, &unused, stream, false);
}
extern "C" int ivx_dev_mc_total(const ivx_mc_params *p, void *scratch_, int64_t *ntris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    const Scratch s = make_scratch(g, p->niso);
    *ntris = 0;
    if (s.nwords == 0) return IVX_OK;
    return mc_read_total(p, s, scratch_, ntris, ivx::S(stream));
}

// Same, with the inside plane (value >= iso[0], source coordinates, the layout of the region-growing planes) handed in
// instead of being derived from the voxels: a resident pipeline that already holds it (ivx_dev_threshold_i16_bits)
// skips the pass over the volume.  One iso-value only.
static int mc_count_bits_impl(const ivx_mc_params *p, const uint64_t *inside_bits, void *scratch_, int64_t *ntris,
                              void *stream, bool ntris_wanted) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    IVX_REQUIRE(p->niso == 1, IVX_EINVAL, "mc_count_bits: one iso-value only");
    const Scratch s = make_scratch(g, p->niso);
    hipStream_t st = ivx::S(stream);
    *ntris = 0;
    if (s.nwords == 0) return IVX_OK;
    IVX_REQUIRE(s.nblocks < 0x7fffffffull && s.nwords < 0xffffffffull, IVX_EINVAL, "mc: piece too large for one launch");
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        g_ext_bits[scratch_] = inside_bits; // read in place until the next count on this scratch
    }
    if ((rc = mc_queue_count(p, g, s, scratch_, st))) return rc;
    return ntris_wanted ? mc_read_total(p, s, scratch_, ntris, st) : IVX_OK;
}
extern "C" int ivx_dev_mc_count_bits(const ivx_mc_params *p, const uint64_t *inside_bits, void *scratch_, int64_t *ntris,
                                     void *stream) {
    return mc_count_bits_impl(p, inside_bits, scratch_, ntris, stream, true);
}
extern "C" int ivx_dev_mc_count_bits_async(const ivx_mc_params *p, const uint64_t *inside_bits, void *scratch_, void *stream) {
    int64_t unused = 0;
    return mc_count_bits_impl(p, inside_bits, scratch_, &unused, stream, false);
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

// ivx_dev_mc_emit for a uint8 mask whose values are KNOWN to be v_out outside the inside plane of the count, v_sel where
// `sel_bits` (same layout as that plane) has a bit and v_in elsewhere inside: no voxel is read, the soup is the same.
extern "C" int ivx_dev_mc_emit_levels(const ivx_mc_params *p, const void *scratch, const uint64_t *sel_bits, double v_out,
                                      double v_in, double v_sel, float *tris, int64_t max_tris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    IVX_REQUIRE(p->dtype == IVX_U8 && p->niso == 1 && sel_bits, IVX_EINVAL, "mc_emit_levels: uint8 mask, one iso-value");
    // (every OUTSIDE end of an edge, the virtual padding included, is interpolated as v_out: a pad value that is merely below
    // the iso-value would give silently different border vertices, so it has to BE v_out -- as the indexed variant requires)
    IVX_REQUIRE(v_out < p->iso[0] && v_in >= p->iso[0] && v_sel >= p->iso[0] && (double)p->pad_value == v_out, IVX_EINVAL,
                "mc_emit_levels: v_out must lie below the iso-value and equal the padding value, v_in and v_sel at or above it");
    const Scratch s = make_scratch(g, p->niso);
    if (s.nwords == 0 || max_tris <= 0) return IVX_OK;
    const McLevels lv = make_levels(sel_bits, p->iso[0], v_out, v_in, v_sel);
    return run_emit<uint8_t>(p, g, s, nullptr, (const char *)scratch, tris, max_tris, ivx::S(stream), &lv);
}

// The surface in one launch (k_mc_fused): no per-word counts, no scan launch, no triangle list.  One iso-value.  `inside_bits`
// = the plane "value >= iso[0]" in source coordinates when the caller holds it (a resident pipeline's threshold pass), else
// NULL: it is derived from `a` into the scratch first.  At most max_tris triangles are written; ivx_dev_mc_total (same params,
// scratch, stream) then returns how many there ARE -- a caller whose buffer was too small calls again with a larger one.  The
// soup is the one ivx_dev_mc_count + ivx_dev_mc_emit produce, bit for bit and in the same order.  The per-word counts in the
// scratch are NOT produced: the indexed-mesh calls still need their ivx_dev_mc_count first.
template <typename T>
static int run_fused(const ivx_mc_params *p, const Geom &g, const Scratch &s, const void *a, const uint64_t *bits, char *scratch,
                     float *tris, int64_t max_tris, hipStream_t st, const McLevels *lv) {
    uint64_t *boff = (uint64_t *)(scratch + s.off_boff);
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        g_list_built.erase(scratch); // whatever list was built from this scratch's old counts is void
    }
    IVX_HIP(hipMemsetAsync(boff, 0, (s.nblocks + 1) * 8, st)); // the look-back's status words + the total
    uint32_t *ticket = (uint32_t *)(scratch + s.off_bsum);         // (the workgroup sums of the four-launch path: unused here)
    IVX_HIP(hipMemsetAsync(ticket, 0, 4, st));
    IVX_REQUIRE((uint64_t)s.nwords * 64u * MC_MAX_TRI < (1ull << 40), IVX_EINVAL, "mc_surface: more triangles than the look-back's 40-bit prefix holds");
    if (lv)
        hipLaunchKernelGGL((k_mc_fused<T, true>), dim3((unsigned)s.nblocks), dim3(256), 0, st, bits, (const T *)a, g, s.nwords, pad_bits(p, 0),
                           p->iso[0], (unsigned long long *)boff, ticket, boff + s.nblocks, (uint64_t)(max_tris > 0 ? max_tris : 0), tris, *lv);
    else
        hipLaunchKernelGGL((k_mc_fused<T, false>), dim3((unsigned)s.nblocks), dim3(256), 0, st, bits, (const T *)a, g, s.nwords, pad_bits(p, 0),
                           p->iso[0], (unsigned long long *)boff, ticket, boff + s.nblocks, (uint64_t)(max_tris > 0 ? max_tris : 0), tris,
                           McLevels{nullptr, 0.0, 0.0, 0.0, {0.0, 0.0, 0.0, 0.0}});
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_mc_surface(const ivx_mc_params *p, const void *a, const uint64_t *inside_bits, void *scratch_, float *tris,
                                  int64_t max_tris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    IVX_REQUIRE(p->niso == 1, IVX_EINVAL, "mc_surface: one iso-value only (two-iso pieces take ivx_dev_mc_count + ivx_dev_mc_emit)");
    IVX_REQUIRE(a && scratch_ && (tris || max_tris <= 0), IVX_EINVAL, "mc_surface: null buffer");
    const Scratch s = make_scratch(g, p->niso);
    if (s.nwords == 0) return IVX_OK;
    IVX_REQUIRE(s.nblocks < 0x7fffffffull && s.nwords < 0xffffffffull, IVX_EINVAL, "mc: piece too large for one launch");
    char *scratch = (char *)scratch_;
    hipStream_t st = ivx::S(stream);
    const uint64_t *bits = inside_bits;
    if (!bits) {
        switch (p->dtype) {
        case IVX_U8: rc = run_bits<uint8_t>(p, g, s, a, (uint8_t *)(scratch + s.off_bits), p->iso[0], p->iso[1], st); break;
        case IVX_I16: rc = run_bits<int16_t>(p, g, s, a, (uint8_t *)(scratch + s.off_bits), p->iso[0], p->iso[1], st); break;
        default: rc = run_bits<uint16_t>(p, g, s, a, (uint8_t *)(scratch + s.off_bits), p->iso[0], p->iso[1], st); break;
        }
        if (rc) return rc;
        bits = (const uint64_t *)(scratch + s.off_bits);
    }
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        if (inside_bits) g_ext_bits[scratch_] = inside_bits;
        else g_ext_bits.erase(scratch_);
    }
    switch (p->dtype) {
    case IVX_U8: return run_fused<uint8_t>(p, g, s, a, bits, scratch, tris, max_tris, st, nullptr);
    case IVX_I16: return run_fused<int16_t>(p, g, s, a, bits, scratch, tris, max_tris, st, nullptr);
    default: return run_fused<uint16_t>(p, g, s, a, bits, scratch, tris, max_tris, st, nullptr);
    }
}

// ... for a uint8 mask whose bytes are KNOWN through two planes (see ivx_dev_mc_emit_levels): v_out outside `inside_bits`, v_sel
// where `sel_bits` has a bit, v_in elsewhere inside.  No voxel is read.
extern "C" int ivx_dev_mc_surface_levels(const ivx_mc_params *p, const uint64_t *inside_bits, const uint64_t *sel_bits, double v_out,
                                         double v_in, double v_sel, void *scratch_, float *tris, int64_t max_tris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    IVX_REQUIRE(p->dtype == IVX_U8 && p->niso == 1 && inside_bits && sel_bits && scratch_, IVX_EINVAL,
                "mc_surface_levels: uint8 mask, one iso-value, both planes");
    IVX_REQUIRE(v_out < p->iso[0] && v_in >= p->iso[0] && v_sel >= p->iso[0] && (double)p->pad_value == v_out, IVX_EINVAL,
                "mc_surface_levels: v_out must lie below the iso-value and equal the padding value, v_in and v_sel at or above it");
    const Scratch s = make_scratch(g, p->niso);
    if (s.nwords == 0) return IVX_OK;
    IVX_REQUIRE(s.nblocks < 0x7fffffffull && s.nwords < 0xffffffffull, IVX_EINVAL, "mc: piece too large for one launch");
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        g_ext_bits[scratch_] = inside_bits;
    }
    const McLevels lv = make_levels(sel_bits, p->iso[0], v_out, v_in, v_sel);
    return run_fused<uint8_t>(p, g, s, nullptr, inside_bits, (char *)scratch_, tris, max_tris, ivx::S(stream), &lv);
}

// The list pass of ivx_dev_mc_emit on its own: it needs the counts only, not the voxels, so a pipeline can queue it (on
// the stream the emit will use: the list lives in that stream's workspace) before the values the emit interpolates are
// final.  The emit that follows with max_tris <= this max_tris skips its own list pass.
extern "C" int ivx_dev_mc_list(const ivx_mc_params *p, const void *scratch, int64_t max_tris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    const Scratch s = make_scratch(g, p->niso);
    if (s.nwords == 0 || s.nblocks == 0 || max_tris <= 0) return IVX_OK;
    IVX_REQUIRE(s.nwords < 0xffffffffull, IVX_EINVAL, "mc: piece too large for 32-bit cell-word ids");
    hipStream_t st = ivx::S(stream);
    void *d_list;
    if ((rc = ivx::ws_get_s(ivx::WS_MCLIST, st, (size_t)max_tris * 8 + 64, &d_list))) return rc;
    if ((rc = run_list(p, g, s, (const char *)scratch, d_list, max_tris, st))) return rc;
    std::lock_guard<std::mutex> lk(g_split_mu);
    g_list_built[scratch] = ListBuilt{d_list, max_tris};
    g_list_owner[d_list] = scratch;
    return IVX_OK;
}

extern "C" int ivx_marching_cubes(const ivx_mc_params *p, const void *a, const int64_t strides[3], float *tris,
                                  int64_t max_tris, int64_t *ntris) {
    ivx::HostCallGuard host_guard__;
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
    // one iso-value (from_binary pieces): the single-launch surface, first with no room at all -- a pure count --, then with the
    // buffer the count asks for (the inside plane of the first pass is still in the scratch) -- opt-in, IVX_MC_ONE_LAUNCH=1:
    // measured slower than count + list + emit at 512^3 (see k_mc_fused).
    const char *e = getenv("IVX_MC_ONE_LAUNCH");
    const bool one = p->niso == 1 && e && e[0] == '1';
    if (one) {
        if ((rc = ivx_dev_mc_surface(p, d_a, nullptr, d_scr, nullptr, 0, nullptr))) return rc;
        if ((rc = ivx_dev_mc_total(p, d_scr, &cnt, nullptr))) return rc;
    } else if ((rc = ivx_dev_mc_count(p, d_a, d_scr, &cnt, nullptr)))
        return rc;
    *ntris = cnt;
    if (!tris || cnt == 0) return IVX_OK;
    IVX_REQUIRE(max_tris >= cnt, IVX_ERANGE, "mc: output buffer holds %lld triangles, %lld needed", (long long)max_tris,
                (long long)cnt);
    void *d_tris;
    if ((rc = ws_get(WS_OUT, (size_t)cnt * 36, &d_tris))) return rc;
    if (one) {
        const Scratch s = make_scratch(g, p->niso);
        if ((rc = ivx_dev_mc_surface(p, d_a, (const uint64_t *)((const char *)d_scr + s.off_bits), d_scr, (float *)d_tris, cnt, nullptr))) return rc;
    } else if ((rc = ivx_dev_mc_emit(p, d_a, d_scr, (float *)d_tris, cnt, nullptr)))
        return rc;
    IVX_HIP(hipMemcpy(tris, d_tris, (size_t)cnt * 36, hipMemcpyDeviceToHost));
    return IVX_OK;
}

// The host form in two halves that share the device work: _begin uploads the piece once, counts AND emits into the library's
// output block and returns the count; _fetch -- the very next host-level call -- copies the soup into the array the caller has
// sized from that count.  (ivx_marching_cubes' count call + emit call upload the piece twice and count twice: 2.6 + 12 ms for a
// 512^3 mask whose kernels take 0.15 ms -- VERDICT r5 weak #11.)  _fetch refuses (IVX_EINVAL) when another host-level call came
// in between: the caller then takes the two-call form.
static uint64_t g_mc_begin_epoch = 0;
static int64_t g_mc_begin_count = 0;
extern "C" int ivx_marching_cubes_begin(const ivx_mc_params *p, const void *a, const int64_t strides[3], int64_t *ntris) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(p && a && ntris, IVX_EINVAL, "marching_cubes_begin: null argument");
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    g_mc_begin_epoch = 0;
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
    if (cnt) {
        void *d_tris;
        if ((rc = ws_get(WS_OUT, (size_t)cnt * 36, &d_tris))) return rc;
        if ((rc = ivx_dev_mc_emit(p, d_a, d_scr, (float *)d_tris, cnt, nullptr))) return rc;
    }
    g_mc_begin_epoch = host_epoch();
    g_mc_begin_count = cnt;
    return IVX_OK;
}
extern "C" int ivx_marching_cubes_fetch(float *tris, int64_t ntris) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(g_mc_begin_epoch != 0 && host_epoch() == g_mc_begin_epoch + 1 && ntris == g_mc_begin_count, IVX_EINVAL,
                "marching_cubes_fetch: not the call right behind ivx_marching_cubes_begin (or another count): take ivx_marching_cubes");
    g_mc_begin_epoch = 0;
    if (ntris == 0) return IVX_OK;
    IVX_REQUIRE(tris, IVX_EINVAL, "marching_cubes_fetch: null buffer");
    void *d_tris;
    int rc;
    if ((rc = ws_get(WS_OUT, (size_t)ntris * 36, &d_tris))) return rc; // (the block _begin filled: large enough, not reallocated)
    return copy_d2h(tris, d_tris, (size_t)ntris * 36); // (the page-locked lanes for destinations whose pages are not resident yet)
}

// ---- indexed mesh API: must follow ivx_dev_mc_count on the same params / scratch / stream --------------------------
static int mc_indexed_count_impl(const ivx_mc_params *p, const void *a, const void *scratch_, int64_t *nverts, void *stream,
                                 bool levels);
extern "C" int ivx_dev_mc_indexed_count(const ivx_mc_params *p, const void *a, const void *scratch_, int64_t *nverts,
                                        void *stream) {
    return mc_indexed_count_impl(p, a, scratch_, nverts, stream, false);
}
// ivx_dev_mc_indexed_count for a mask whose bytes are known to lie strictly on either side of the iso-value (the levels of
// ivx_dev_mc_emit_levels; follows ivx_dev_mc_count_bits): "value > iso" IS the inside plane, so the pass over the mask that
// derives the strictly-inside plane is a copy of 1/8 byte per voxel instead of a read of the volume.
extern "C" int ivx_dev_mc_indexed_count_levels(const ivx_mc_params *p, const void *scratch_, int64_t *nverts, void *stream) {
    IVX_REQUIRE(p && p->dtype == IVX_U8 && p->niso == 1, IVX_EINVAL, "mc_indexed_count_levels: uint8 mask, one iso-value");
    return mc_indexed_count_impl(p, nullptr, scratch_, nverts, stream, true);
}
static int mc_indexed_count_impl(const ivx_mc_params *p, const void *a, const void *scratch_, int64_t *nverts, void *stream,
                                 bool levels) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    const Scratch s = make_scratch(g, p->niso);
    *nverts = 0;
    if (s.nwords == 0) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    const MciLayout m = mci_layout(g, s, p->niso);
    void *d_v;
    if ((rc = ivx::ws_get_s(ivx::WS_MCV, st, m.total, &d_v))) return rc;
    // strictly-inside planes: value > iso  <=>  value >= nextafter(iso, +inf)
    const double n0 = std::nextafter(p->iso[0], HUGE_VAL), n1 = std::nextafter(p->iso[1], HUGE_VAL);
    if (levels) {
        IVX_REQUIRE(p->pad_value < p->iso[0], IVX_EINVAL, "mc_indexed_count_levels: the padding must lie below the iso-value");
        IVX_HIP(hipMemcpyAsync(d_v, mc_bits_ptr(scratch_, s, 0), s.bits_words * 8, hipMemcpyDeviceToDevice, st));
    } else {
        switch (p->dtype) {
        case IVX_U8: rc = run_bits<uint8_t>(p, g, s, a, (uint8_t *)d_v, n0, n1, st); break;
        case IVX_I16: rc = run_bits<int16_t>(p, g, s, a, (uint8_t *)d_v, n0, n1, st); break;
        default: rc = run_bits<uint16_t>(p, g, s, a, (uint8_t *)d_v, n0, n1, st); break;
        }
        if (rc) return rc;
    }
    uint32_t tot[2] = {0, 0};
    for (int q = 0; q < p->niso; q++) {
        uint32_t *vbase = (uint32_t *)((char *)d_v + m.off_v + (size_t)q * m.per_iso);
        uint32_t *bsum = vbase + m.npw, *d_total = bsum + m.nsb;
        const uint64_t *bits = mc_bits_ptr(scratch_, s, q);
        const uint64_t *qb = (const uint64_t *)d_v + (size_t)q * s.bits_words;
        const int64_t blocks = ivx::cdiv(m.npw, 256);
        hipLaunchKernelGGL(k_mci_count, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, bits, qb, g,
                           m.npw, pad_bits(p, q), pad_qbits(p, q), vbase, (CrossRec *)((char *)d_v + m.off_rec + (size_t)q * m.per_iso_rec));
        IVX_LAUNCH_CHECK();
        if ((rc = scan_u32_exclusive(vbase, m.npw, bsum, d_total, st))) return rc;
        uint32_t seq;
        if ((rc = ivx::mailbox_publish(d_total, 1, st, &seq))) return rc;
        if ((rc = ivx::mailbox_wait(seq, st, &tot[q], 1))) return rc;
    }
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        g_vsplit[scratch_] = tot[0];
    }
    *nverts = (int64_t)tot[0] + (int64_t)tot[1];
    return IVX_OK;
}

template <typename T>
static int run_indexed(const ivx_mc_params *p, const Geom &g, const Scratch &s, const void *a, const char *scratch,
                       float *verts, int64_t max_verts, int32_t *faces, int64_t max_tris, hipStream_t st,
                       const McLevels *lv = nullptr) {
    IVX_REQUIRE(s.nwords < 0xffffffffull, IVX_EINVAL, "mc: piece too large for 32-bit cell-word ids");
    const MciLayout m = mci_layout(g, s, p->niso);
    void *d_v, *d_list;
    int rc;
    if ((rc = ivx::ws_get_s(ivx::WS_MCV, st, m.total, &d_v))) return rc;
    if ((rc = ivx::ws_get_s(ivx::WS_MCLIST, st, (size_t)max_tris * 8 + 64, &d_list))) return rc;
    const bool have_list = list_ready(scratch, d_list, max_tris); // else: marks the buffer as about to be overwritten
    const uint64_t *boff = (const uint64_t *)(scratch + s.off_boff);
    uint64_t tb[3] = {0, (uint64_t)max_tris, (uint64_t)max_tris};
    uint32_t vsplit = 0;
    {
        std::lock_guard<std::mutex> lk(g_split_mu);
        auto iv = g_vsplit.find(scratch);
        IVX_REQUIRE(iv != g_vsplit.end(), IVX_EINVAL, "mc: ivx_dev_mc_indexed_emit must follow ivx_dev_mc_indexed_count");
        vsplit = iv->second;
        if (p->niso == 2) {
            auto it = g_split.find(scratch);
            IVX_REQUIRE(it != g_split.end(), IVX_EINVAL, "mc: ivx_dev_mc_indexed_emit must follow ivx_dev_mc_count");
            tb[1] = it->second;
        }
    }
    for (int q = 0; q < p->niso; q++) {
        const uint64_t *bits = mc_bits_ptr(scratch, s, q);
        const uint64_t *qb = (const uint64_t *)d_v + (size_t)q * s.bits_words;
        const uint16_t *counts = (const uint16_t *)(scratch + s.off_counts) + (size_t)q * s.nwords;
        const uint32_t *vbase = (const uint32_t *)((const char *)d_v + m.off_v + (size_t)q * m.per_iso);
        const uint32_t id0 = q == 0 ? 0u : vsplit;
        if (!have_list) {
            hipLaunchKernelGGL(k_mc_list, dim3((unsigned)s.nblocks), dim3(256), 0, st, bits, g, s.nwords, pad_bits(p, q), counts,
                               boff + (size_t)q * s.nblocks, (uint64_t *)d_list, (uint64_t)max_tris);
            IVX_LAUNCH_CHECK();
        }
        const int64_t blocks = ivx::cdiv(m.npw, 256);
        const CrossRec *rec = (const CrossRec *)((const char *)d_v + m.off_rec + (size_t)q * m.per_iso_rec);
        if (lv)
            hipLaunchKernelGGL(k_mci_vertices_levels, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, bits, rec,
                               g, m.npw, pad_bits(p, q), *lv, vbase, id0, verts, (uint64_t)max_verts);
        else
            hipLaunchKernelGGL((k_mci_vertices<T>), dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st,
                               (const T *)a, rec, g, m.npw, p->iso[q], vbase, id0, verts, (uint64_t)max_verts);
        IVX_LAUNCH_CHECK();
        const uint64_t first = tb[q], last = tb[q + 1] < (uint64_t)max_tris ? tb[q + 1] : (uint64_t)max_tris;
        if (last > first) {
            hipLaunchKernelGGL(k_mci_faces, dim3((unsigned)ivx::cdiv((int64_t)(last - first), 256)), dim3(256), 0, st, rec, g, vbase,
                               id0, (const uint64_t *)d_list + first, last - first, faces + first * 3);
            IVX_LAUNCH_CHECK();
        }
    }
    return IVX_OK;
}

extern "C" int ivx_dev_mc_indexed_emit(const ivx_mc_params *p, const void *a, const void *scratch, float *verts,
                                       int64_t max_verts, int32_t *faces, int64_t max_tris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    const Scratch s = make_scratch(g, p->niso);
    if (s.nwords == 0 || max_tris <= 0) return IVX_OK;
    IVX_REQUIRE(max_verts < 0x7fffffffll, IVX_EINVAL, "mc: more than 2^31 vertices do not fit int32 face indices");
    hipStream_t st = ivx::S(stream);
    switch (p->dtype) {
    case IVX_U8: return run_indexed<uint8_t>(p, g, s, a, (const char *)scratch, verts, max_verts, faces, max_tris, st);
    case IVX_I16: return run_indexed<int16_t>(p, g, s, a, (const char *)scratch, verts, max_verts, faces, max_tris, st);
    default: return run_indexed<uint16_t>(p, g, s, a, (const char *)scratch, verts, max_verts, faces, max_tris, st);
    }
}

// ivx_dev_mc_indexed_emit after ivx_dev_mc_indexed_count_levels: the vertices from the mask's known byte levels (see
// ivx_dev_mc_emit_levels), no voxel is read; same vertices and faces, bit for bit, as the voxel path gives on that mask.
extern "C" int ivx_dev_mc_indexed_emit_levels(const ivx_mc_params *p, const void *scratch, const uint64_t *sel_bits, double v_out,
                                              double v_in, double v_sel, float *verts, int64_t max_verts, int32_t *faces,
                                              int64_t max_tris, void *stream) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    IVX_REQUIRE(p->dtype == IVX_U8 && p->niso == 1, IVX_EINVAL, "mc_indexed_emit_levels: uint8 mask, one iso-value");
    IVX_REQUIRE(v_out < p->iso[0] && v_in > p->iso[0] && v_sel > p->iso[0] && p->pad_value == v_out, IVX_EINVAL,
                "mc_indexed_emit_levels: v_out (= the padding) must lie below the iso-value, v_in and v_sel above it");
    const Scratch s = make_scratch(g, p->niso);
    if (s.nwords == 0 || max_tris <= 0) return IVX_OK;
    IVX_REQUIRE(max_verts < 0x7fffffffll, IVX_EINVAL, "mc: more than 2^31 vertices do not fit int32 face indices");
    const McLevels lv = make_levels(sel_bits, p->iso[0], v_out, v_in, v_sel);
    return run_indexed<uint8_t>(p, g, s, nullptr, (const char *)scratch, verts, max_verts, faces, max_tris, ivx::S(stream), &lv);
}

// ---- cross-slab stitch API: follows ivx_dev_mc_indexed_emit on the same params / scratch / stream (one iso-value) ------
struct StitchWs {
    PlaneSig *top;     // this piece's top-plane signature (what the rank above receives)
    uint32_t *rmcnt;   // copies dropped per bottom-plane word, then their exclusive scan
    uint32_t *bsum, *total, *gid0;
    size_t bytes;
};
static StitchWs stitch_layout(const Geom &g, uint32_t p0, char *base) {
    StitchWs w;
    const size_t nwp = (size_t)(g.NY * g.WX);
    size_t o = 0;
    auto take = [&](size_t n) { char *q = base ? base + o : nullptr; o += al256(n); return q; };
    w.top = (PlaneSig *)take(nwp * sizeof(PlaneSig));
    w.rmcnt = (uint32_t *)take((nwp + 1) * 4);
    w.bsum = (uint32_t *)take(((size_t)scan_u32_blocks((int64_t)nwp) + 2) * 4);
    w.total = (uint32_t *)take(64);
    w.gid0 = (uint32_t *)take(((size_t)p0 + 1) * 4);
    w.bytes = o;
    return w;
}
static int stitch_ctx(const ivx_mc_params *p, const void *scratch, hipStream_t st, Geom *g, Scratch *s, MciLayout *m, void **d_v) {
    int rc = make_geom(p, g);
    if (rc) return rc;
    IVX_REQUIRE(p->niso == 1, IVX_EINVAL, "mc stitch: one iso-value only");
    *s = make_scratch(*g, p->niso);
    IVX_REQUIRE(s->nwords > 0 && g->NZ >= 2, IVX_EINVAL, "mc stitch: the piece needs at least one cell layer");
    *m = mci_layout(*g, *s, p->niso);
    return ivx::ws_get_s(ivx::WS_MCV, st, m->total, d_v); // (the block ivx_dev_mc_indexed_count filled)
}

extern "C" int ivx_dev_mc_stitch_sig_bytes(const ivx_mc_params *p, size_t *nbytes) {
    Geom g;
    int rc = make_geom(p, &g);
    if (rc) return rc;
    *nbytes = (size_t)(g.NY * g.WX) * sizeof(PlaneSig);
    return IVX_OK;
}

// the signature of this piece's TOP point plane -> `sig` (device, ivx_dev_mc_stitch_sig_bytes): send it to the rank above
extern "C" int ivx_dev_mc_stitch_top_sig(const ivx_mc_params *p, const void *scratch, void *sig, void *stream) {
    Geom g;
    Scratch s;
    MciLayout m;
    void *d_v;
    hipStream_t st = ivx::S(stream);
    int rc = stitch_ctx(p, scratch, st, &g, &s, &m, &d_v);
    if (rc) return rc;
    const uint32_t *vbase = (const uint32_t *)((char *)d_v + m.off_v);
    const int64_t nwp = g.NY * g.WX;
    hipLaunchKernelGGL(k_mci_sig, dim3((unsigned)ivx::cdiv(nwp, 256)), dim3(256), 0, st, mc_bits_ptr(scratch, s, 0),
                       (const uint64_t *)d_v, g, pad_bits(p, 0), pad_qbits(p, 0), vbase, g.NZ - 1, (PlaneSig *)sig);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// this piece's BOTTOM plane against the signature received from the rank below (NULL on the lowest rank):
// vd[0] = nverts, vd[1] = copies this piece drops (two device words: all-gather them over the ranks)
extern "C" int ivx_dev_mc_stitch_match(const ivx_mc_params *p, const void *scratch, const void *nbr_sig, int64_t nverts,
                                       uint32_t *vd, void *stream) {
    Geom g;
    Scratch s;
    MciLayout m;
    void *d_v, *d_w;
    hipStream_t st = ivx::S(stream);
    int rc = stitch_ctx(p, scratch, st, &g, &s, &m, &d_v);
    if (rc) return rc;
    IVX_REQUIRE(nverts >= 0 && nverts < 0x7fffffffll && vd, IVX_EINVAL, "mc stitch: bad vertex count");
    const int64_t nwp = g.NY * g.WX;
    StitchWs w = stitch_layout(g, 0, nullptr);
    if ((rc = ivx::ws_get_s(ivx::WS_MCST, st, w.bytes + ((size_t)nverts + 1) * 4 + 256, &d_w))) return rc;
    w = stitch_layout(g, (uint32_t)nverts, (char *)d_w); // (gid0 sized for the worst case: every vertex in the bottom plane)
    IVX_HIP(hipMemsetAsync(vd, 0, 8, st));
    hipLaunchKernelGGL(k_mci_match, dim3((unsigned)ivx::cdiv(nwp, 256)), dim3(256), 0, st, mc_bits_ptr(scratch, s, 0),
                       (const uint64_t *)d_v, g, pad_bits(p, 0), pad_qbits(p, 0), (const PlaneSig *)nbr_sig, w.rmcnt, vd,
                       (uint32_t)nverts);
    IVX_LAUNCH_CHECK();
    return scan_u32_exclusive(w.rmcnt, nwp, w.bsum, w.total, st);
}

// vd_all = the (nverts, dropped) pairs of every rank in rank order (device, 2 * world words).  `faces` (ntris x 3 local
// ids) become global ids in place; the vertices this piece keeps are written to `verts_out` in their old order
// (nverts - dropped of them; global id of the first one = sum over the ranks below of nverts - dropped).
extern "C" int ivx_dev_mc_stitch_apply(const ivx_mc_params *p, const void *scratch, const void *nbr_sig, const uint32_t *vd_all,
                                       int rank, const float *verts, int64_t nverts, int32_t *faces, int64_t ntris,
                                       float *verts_out, void *stream) {
    Geom g;
    Scratch s;
    MciLayout m;
    void *d_v, *d_w;
    hipStream_t st = ivx::S(stream);
    int rc = stitch_ctx(p, scratch, st, &g, &s, &m, &d_v);
    if (rc) return rc;
    IVX_REQUIRE(rank >= 0 && vd_all && nverts >= 0 && nverts < 0x7fffffffll, IVX_EINVAL, "mc stitch: bad arguments");
    const int64_t nwp = g.NY * g.WX;
    StitchWs w = stitch_layout(g, 0, nullptr);
    if ((rc = ivx::ws_get_s(ivx::WS_MCST, st, w.bytes + ((size_t)nverts + 1) * 4 + 256, &d_w))) return rc;
    w = stitch_layout(g, (uint32_t)nverts, (char *)d_w);
    const uint32_t *vbase = (const uint32_t *)((char *)d_v + m.off_v);
    // first local id above the bottom plane's words: read through the mailbox (sizes nothing, but the kernels need it)
    uint32_t p0 = (uint32_t)nverts;
    if (g.NZ > 1) {
        uint32_t seq;
        if ((rc = ivx::mailbox_publish(vbase + nwp, 1, st, &seq))) return rc;
        if ((rc = ivx::mailbox_wait(seq, st, &p0, 1))) return rc;
    }
    hipLaunchKernelGGL(k_mci_gid0, dim3((unsigned)ivx::cdiv(nwp, 256)), dim3(256), 0, st, mc_bits_ptr(scratch, s, 0),
                       (const uint64_t *)d_v, g, pad_bits(p, 0), pad_qbits(p, 0), vbase, (const PlaneSig *)nbr_sig, w.rmcnt,
                       vd_all, rank, w.gid0);
    IVX_LAUNCH_CHECK();
    if (ntris > 0) {
        const int64_t n3 = ntris * 3, blocks = ivx::cdiv(n3, 256);
        hipLaunchKernelGGL(k_mci_stitch_faces, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, faces, n3, p0,
                           w.gid0, vd_all, rank);
        IVX_LAUNCH_CHECK();
    }
    if (nverts > 0) {
        const int64_t blocks = ivx::cdiv(nverts, 256);
        hipLaunchKernelGGL(k_mci_stitch_verts, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, verts, nverts,
                           p0, w.gid0, vd_all, rank, verts_out);
        IVX_LAUNCH_CHECK();
    }
    return IVX_OK;
}

// Host form: strided piece in, indexed mesh out.  verts == NULL -> counts only (*nverts, *ntris).
extern "C" int ivx_marching_cubes_indexed(const ivx_mc_params *p, const void *a, const int64_t strides[3], float *verts,
                                          int64_t max_verts, int32_t *faces, int64_t max_tris, int64_t *nverts,
                                          int64_t *ntris) {
    ivx::HostCallGuard host_guard__;
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
    int64_t nt = 0, nv = 0;
    if ((rc = ivx_dev_mc_count(p, d_a, d_scr, &nt, nullptr))) return rc;
    if ((rc = ivx_dev_mc_indexed_count(p, d_a, d_scr, &nv, nullptr))) return rc;
    *ntris = nt;
    *nverts = nv;
    if (!verts || !faces || nt == 0) return IVX_OK;
    IVX_REQUIRE(max_tris >= nt && max_verts >= nv, IVX_ERANGE, "mc: output buffers too small (%lld verts, %lld triangles needed)",
                (long long)nv, (long long)nt);
    void *d_verts, *d_faces;
    if ((rc = ws_get(WS_OUT, (size_t)nv * 12 + 64, &d_verts))) return rc;
    if ((rc = ws_get(WS_AUX1, (size_t)nt * 12 + 64, &d_faces))) return rc;
    if ((rc = ivx_dev_mc_indexed_emit(p, d_a, d_scr, (float *)d_verts, nv, (int32_t *)d_faces, nt, nullptr))) return rc;
    IVX_HIP(hipMemcpy(verts, d_verts, (size_t)nv * 12, hipMemcpyDeviceToHost));
    IVX_HIP(hipMemcpy(faces, d_faces, (size_t)nt * 12, hipMemcpyDeviceToHost));
    return IVX_OK;
}
