// k_wsift.hip -- the marker flood of do_watershed's IFT branch on the GPU.
//
// Replaces scipy.ndimage.watershed_ift as called by invesalius/data/watershed_process.py:44-46,54-57 (3-D) and
// invesalius/data/styles.py:1958-1983 (one slice).  scipy's routine (ndimage/src/ni_measure.c, NI_WatershedIFT) is a
// strictly serial bucket-queue flood; what it computes can be said without the queue:
//
//   cost   C(p) = min over paths from a marker of the largest arc |I(a) - I(b)| on the path (unique);
//   ENTRY  a voxel of cost c that has a neighbour v with C(v) < c and |I(v) - I(p)| == c (it is queued at cost c before
//          level c starts), plus every marker voxel (level 0);
//   ZONE   a connected set of non-entry voxels of equal cost c joined by arcs <= c: the first entry next to it that the
//          serial loop pops claims ALL of it (the queue is a stack: what a pop pushes is popped next);
//   ORDER  the stack pops, at level c, the entry pushed last = the one whose parent (its earliest-popped admissible
//          lower-cost neighbour) was popped latest.  Pop times are only ever compared between voxels of DIFFERENT labels,
//          and everything a popped entry claims is popped in one uninterrupted stretch, so a coarse time stamp per voxel
//          is enough:  tau(voxel of level c) = base_c + class of its key, key(entry) = min tau over its admissible
//          parents, classes = runs of consecutive keys (descending) that carry the same label.  Level 0: key = raster
//          rank of the marker voxel (markers are queued in raster order, the last one is popped first).
//
// Pipeline (all on one stream, no CPU arithmetic):
//   1. k_ws_relax      chaotic min-max relaxation of C over 16x16x8 tiles staged in LDS with their 1-voxel halo, only
//                      dirty tiles per round (compact list), rounds until nothing changes;
//   2. k_ws_entries    entry flags; k_ws_runs / k_ws_union / k_ws_flatten: zones by union-find (x-runs first);
//   3. k_ws_hist / k_ws_scatter: entries bucketed by level (markers first, in raster order);
//   4. per non-empty level, ascending: k_ws_keys (key per entry, used-key bitmap) -> k_ws_rank (one workgroup: classes
//      of the used keys, new time stamps, their labels) -> k_ws_claim (entries take their stamp and atomicMin it into
//      the zones they touch);
//   5. k_ws_labels     label = label of the voxel's (or its zone's) time stamp.
//
// Neighbours are taken by LINEAR index like scipy does (its extent test only rejects indices outside [0, size), so the
// last voxel of a row is a neighbour of the first voxel of the next one): every tile cell is addressed by
// z*HW + y*W + x with x in [-1, W], y in [-1, H], which lands on exactly those wrap-around voxels.
//
// Parity: bit-identical to the defect-free statement of scipy's algorithm (oracle/ivx_oracle_wsz.c).  scipy's C source
// has a linked-list defect (`if (p->next || p->prev)` misses the only element of a bucket) that, on inputs where it
// fires, processes some voxels late or never; those inputs differ from live scipy in a handful of voxels (counted by
// tests/test_gpu_wsift.py, reported by bench.py).  Positive markers only (the reference passes 0 / 1 / 2).
#include <algorithm>
#include <vector>

#include "ivx_internal.h"
#include "scan_u32.h"

namespace {
using namespace ivx;

// tile: TX * TY == 256 lanes, one z-column of TZ voxels per lane (32x8x8: 39.5 ms, 16x16x8: 36.4, 16x16x16: 35.5 at 512^3)
#ifndef IVX_WS_TX
#define IVX_WS_TX 16
#define IVX_WS_TY 16
#endif
constexpr int TX = IVX_WS_TX, TY = IVX_WS_TY, TZ = 8, BX = TX + 2, BY = TY + 2, BZ = TZ + 2, NCELL = BX * BY * BZ;
constexpr uint32_t ENTRY = 0xFFFFFFFFu, NONE = 0xFFFFFFFFu, CINF = 0xFFFFu;
constexpr int RELAX_ITCAP = 64;
constexpr int32_t NOLAB = 0;

struct WsGeom {
    int64_t d, h, w, hw, n;
    int ntx, nty, ntz;
    int64_t ntiles;
    uint32_t smask; // bit k = (dz+1)*9 + (dy+1)*3 + (dx+1) of the 3x3x3 structure, centre cleared
};

struct WsState {
    uint32_t base;     // next free time stamp
    uint32_t overflow; // time stamps ran out of the table
    uint32_t neg;      // a negative marker was seen
    uint32_t pad0;
    uint32_t nlist;    // dirty tiles of the next round            } read by the host
    uint32_t minrej;   // smallest cost refused by the gate so far } after every round
    uint32_t assigned; // voxels that have a finite cost           } (one mailbox message)
    uint32_t sweeps;   // LDS sweeps over all tile visits (statistics)
};

template <int CONN> __device__ __forceinline__ bool has_off(uint32_t smask, int k) {
    if (CONN == 26) return k != 13;
    const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
    const int m = (dz != 0) + (dy != 0) + (dx != 0);
    if (CONN == 6) return m == 1;
    if (CONN == 18) return m == 1 || m == 2;
    return (smask >> k) & 1u;
}

__device__ __forceinline__ uint32_t absdiff(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }

// lattice coordinates (x in [-1, W], y, z anything) -> owning tile, or -1 outside the volume
__device__ __forceinline__ int64_t owner_tile(const WsGeom &g, int64_t z, int64_t y, int64_t x) {
    while (x < 0) { x += g.w; y -= 1; }
    while (x >= g.w) { x -= g.w; y += 1; }
    while (y < 0) { y += g.h; z -= 1; }
    while (y >= g.h) { y -= g.h; z += 1; }
    if (z < 0 || z >= g.d) return -1;
    return ((z / TZ) * g.nty + y / TY) * g.ntx + x / TX;
}

__device__ __forceinline__ void tile_origin(const WsGeom &g, int64_t tile, int &z0, int &y0, int &x0) {
    const int tx = (int)(tile % g.ntx);
    const int64_t r = tile / g.ntx;
    x0 = tx * TX;
    y0 = (int)(r % g.nty) * TY;
    z0 = (int)(r / g.nty) * TZ;
}

// stage the tile and its halo: cell = cost << 16 | intensity; cells outside [0, n) can never lower anything.
// Rows are 16 voxels + 2 halo cells.  When W is a multiple of 8 the 16 interior cells of every lattice row start on a
// 16-byte boundary (the wrap-around rows too: everything is addressed by linear index), so a lane fetches 8 cells of
// both arrays with two 16-byte loads (720 per tile instead of 6800 two-byte loads); the 360 halo cells, and any row
// that straddles the ends of the volume, go cell by cell.  Staging was 40 % of a visit's time before this.
__device__ __forceinline__ void load_tile(const WsGeom &g, int z0, int y0, int x0, const uint16_t *__restrict__ I,
                                          const uint16_t *C, uint32_t *s) {
    static_assert(TX % 8 == 0, "rows of whole 8-cell chunks");
    if ((g.w & 7) == 0) {
        constexpr int NROW = BZ * BY, CH = TX / 8, NITEM = NROW * CH; // (row, chunk of 8 cells)
#pragma unroll
        for (int q = 0; q < (NITEM + 255) / 256; q++) {
            const int it = threadIdx.x + q * 256;
            if (it < NITEM) {
                const int row = it / CH, half = it % CH, ly = row % BY, lz = row / BY;
                const int64_t L = (int64_t)(z0 + lz - 1) * g.hw + (int64_t)(y0 + ly - 1) * g.w + x0 + half * 8;
                uint32_t *dst = s + row * BX + 1 + half * 8;
                if (L >= 0 && L + 8 <= g.n) {
                    const uint4 cv = *reinterpret_cast<const uint4 *>(C + L);
                    const uint4 iv = *reinterpret_cast<const uint4 *>(I + L);
                    const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, iw[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        dst[2 * e] = (cw[e] << 16) | (iw[e] & 0xFFFFu);
                        dst[2 * e + 1] = (cw[e] & 0xFFFF0000u) | (iw[e] >> 16);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int64_t Le = L + e;
                        dst[e] = (Le >= 0 && Le < g.n) ? ((uint32_t)C[Le] << 16) | I[Le] : CINF << 16;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < (NROW * 2 + 255) / 256; q++) { // the two x-halo cells of every row
            const int it = threadIdx.x + q * 256;
            if (it < NROW * 2) {
                const int row = it >> 1, side = it & 1, ly = row % BY, lz = row / BY;
                const int64_t L = (int64_t)(z0 + lz - 1) * g.hw + (int64_t)(y0 + ly - 1) * g.w + x0 + (side ? TX : -1);
                s[row * BX + (side ? BX - 1 : 0)] = (L >= 0 && L < g.n) ? ((uint32_t)C[L] << 16) | I[L] : CINF << 16;
            }
        }
        return;
    }
    constexpr int PER = (NCELL + 255) / 256;
    uint32_t cv[PER], iv[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int c = threadIdx.x + q * 256;
        const int lx = c % BX, ly = (c / BX) % BY, lz = c / (BX * BY);
        const int64_t L = (int64_t)(z0 + lz - 1) * g.hw + (int64_t)(y0 + ly - 1) * g.w + (x0 + lx - 1);
        const bool ok = c < NCELL && L >= 0 && L < g.n;
        const int64_t La = ok ? L : 0;
        cv[q] = ok ? (uint32_t)C[La] : CINF;
        iv[q] = ok ? (uint32_t)I[La] : 0u;
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int c = threadIdx.x + q * 256;
        if (c < NCELL) s[c] = (cv[q] << 16) | iv[q];
    }
}

template <typename MT>
__global__ __launch_bounds__(256) void k_ws_init(WsGeom g, const MT *__restrict__ mk, uint16_t *__restrict__ C,
                                                 uint8_t *__restrict__ dirty, uint32_t *__restrict__ bcount, WsState *st) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * 2048;
    uint32_t mine = 0;
    bool neg = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int64_t p = b0 + j * 256 + threadIdx.x;
        if (p >= g.n) continue;
        const int m = (int)mk[p];
        C[p] = m ? (uint16_t)0 : (uint16_t)CINF;
        if (m) {
            mine++;
            neg |= m < 0;
            const int64_t z = p / g.hw, r = p - z * g.hw, y = r / g.w, x = r - y * g.w;
            dirty[((z / TZ) * g.nty + y / TY) * g.ntx + x / TX] = 1;
        }
    }
    if (mine) atomicAdd(&s_cnt, mine);
    if (neg) st->neg = 1;
    __syncthreads();
    if (threadIdx.x == 0) bcount[blockIdx.x] = s_cnt;
}

// markers in raster order -> elist[0 .. M), keys = their raster rank, labels of the ranks
template <typename MT>
__global__ __launch_bounds__(256) void k_ws_marker_list(WsGeom g, const MT *__restrict__ mk, const uint32_t *__restrict__ boff,
                                                        uint32_t *__restrict__ elist, uint32_t *__restrict__ key,
                                                        int32_t *__restrict__ lab) {
    __shared__ uint32_t s_c[8][4];
    const int64_t b0 = (int64_t)blockIdx.x * 2048;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int m[8];
    uint32_t pre[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int64_t p = b0 + j * 256 + threadIdx.x;
        m[j] = p < g.n ? (int)mk[p] : 0;
        const unsigned long long b = __ballot(m[j] != 0);
        pre[j] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_c[j][wv] = __popcll(b);
    }
    __syncthreads();
    const uint32_t base = boff[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (!m[j]) continue;
        uint32_t off = base + pre[j];
        for (int q = 0; q < j * 4 + wv; q++) off += s_c[q >> 2][q & 3];
        elist[off] = (uint32_t)(b0 + j * 256 + threadIdx.x);
        key[off] = off;
        lab[off] = m[j];
    }
}

__global__ __launch_bounds__(256) void k_ws_fill_used(uint32_t *used, uint32_t m) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    const uint32_t nw = (m + 31) >> 5;
    if (w >= nw) return;
    used[w] = (w == nw - 1 && (m & 31)) ? ((1u << (m & 31)) - 1u) : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void k_ws_build_list(int64_t ntiles, uint8_t *__restrict__ dirty, uint32_t *__restrict__ list,
                                                       WsState *st) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool d = t < ntiles && dirty[t];
    const unsigned long long b = __ballot(d);
    if (!b) return;
    const int lane = threadIdx.x & 63;
    uint32_t off = 0;
    if (lane == 0) off = atomicAdd(&st->nlist, (uint32_t)__popcll(b));
    off = __shfl(off, 0, 64);
    if (d) {
        list[off + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)t;
        dirty[t] = 0;
    }
}

// one visit of a dirty tile: relax to the local fix-point, write the changed costs back, wake the tiles that read them.
// The first sweep evaluates every voxel (lane = z-column); after that only voxels a neighbour of which changed are
// looked at again -- a flag word per z-column in LDS -- and those few are pooled into a work queue in LDS and dealt out
// one per lane (sparse sweeps walked column by column kept one or two lanes of a wave busy: 1.6e9 of the 3.7e9 voxel
// evaluations of a 512^3 flood, at a tenth of the lanes).
// theta gates the flood: a cost above it is not accepted yet (the tile is parked in `pending`), so that below the level
// where the bulk of the volume connects only final costs spread -- no wave of provisional costs to correct later.
template <int CONN, bool LDS_CHG>
__device__ __forceinline__ bool ws_eval(uint32_t *s, uint32_t (*s_act)[TX], uint32_t (*s_chg)[TX], int lx, int ly, int zz, int nz,
                                        uint32_t smask, uint32_t theta, uint32_t &fresh, uint32_t &rej) {
    const int ci = ((zz + 1) * BY + (ly + 1)) * BX + (lx + 1);
    const uint32_t cell = s[ci];
    const uint32_t c = cell >> 16, iv = cell & 0xFFFFu;
    if (c == 0) return false;
    uint32_t best = c;
#pragma unroll
    for (int k = 0; k < 27; k++) {
        if (!has_off<CONN>(smask, k)) continue;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const uint32_t qv = s[ci + (dz * BY + dy) * BX + dx];
        const uint32_t m = max(qv >> 16, absdiff(qv & 0xFFFFu, iv));
        best = min(best, m);
    }
    if (best >= c) return false;
    if (best > theta) {
        rej = min(rej, best);
        return false;
    }
    s[ci] = (best << 16) | iv;
    if (LDS_CHG) atomicOr(&s_chg[ly][lx], 1u << zz); // pooled sweeps: any lane may change any voxel
    fresh += c == CINF;
    // the neighbours inside the tile have to look again
#pragma unroll
    for (int cy = -1; cy <= 1; cy++) {
#pragma unroll
        for (int cx = -1; cx <= 1; cx++) {
            uint32_t m3 = 0;
#pragma unroll
            for (int dz = -1; dz <= 1; dz++)
                if (has_off<CONN>(smask, (dz + 1) * 9 + (cy + 1) * 3 + (cx + 1))) m3 |= 1u << (dz + 1);
            if (!m3) continue;
            const int tx = lx + cx, ty = ly + cy;
            if ((unsigned)tx >= (unsigned)TX || (unsigned)ty >= (unsigned)TY) continue;
            const uint32_t bits = ((m3 << zz) >> 1) & ((1u << nz) - 1u);
            if (bits) atomicOr(&s_act[ty][tx], bits);
        }
    }
    return true;
}

template <int CONN>
__global__ __launch_bounds__(256) void k_ws_relax(WsGeom g, const uint16_t *__restrict__ I, uint16_t *C,
                                                  const uint32_t *__restrict__ list, uint8_t *dirty, uint8_t *pending,
                                                  WsState *st, uint32_t theta_flags) {
    __shared__ uint32_t s[NCELL];
    __shared__ uint32_t s_act[TY][TX], s_chg[TY][TX];
    __shared__ uint16_t s_queue[CONN == 6 ? 2 : TX * TY * TZ]; // (the 6-neighbour form never pools)
    __shared__ uint32_t s_qn[2];
    __shared__ uint32_t s_new, s_rej, s_ev2;
    const int64_t tile = list[blockIdx.x];
    const uint32_t theta = theta_flags & 0xFFFFu; // bit 31 of the argument: collect the sweep statistic
    int z0, y0, x0;
    tile_origin(g, tile, z0, y0, x0);
    load_tile(g, z0, y0, x0, I, C, s);
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    s_act[ly][lx] = 0;
    s_chg[ly][lx] = 0;
    if (threadIdx.x == 0) { s_new = 0; s_rej = NONE; s_ev2 = 0; s_qn[0] = 0; s_qn[1] = 0; }
    __syncthreads();
    const bool col = x0 + lx < g.w && y0 + ly < g.h;
    const int nz = min(TZ, (int)(g.d - z0));
    uint32_t fresh = 0, rej = NONE, chg = 0; // chg: changes this lane made to ITS column (a register is enough there)
    // sweep 0: every voxel, column by column
    if (col) {
        for (int zz = 0; zz < nz; zz++)
            if (ws_eval<CONN, false>(s, s_act, s_chg, lx, ly, zz, nz, g.smask, theta, fresh, rej)) chg |= 1u << zz;
    }
    __syncthreads();
    int it = 1;
    bool more = true;
    if (CONN == 6) {
        // six cheap neighbours per voxel: walking the flagged voxels of its own column costs a lane less than the two
        // barriers of the pooled form (measured 36.6 vs 42.3 ms at 512^3)
        while (more && it < RELAX_ITCAP) {
            bool any = false;
            uint32_t a = col ? atomicExch(&s_act[ly][lx], 0u) : 0u;
            while (a) {
                const int zz = (it & 1) ? 31 - __clz(a) : __ffs(a) - 1; // alternate the sweep direction
                a &= ~(1u << zz);
                if (ws_eval<CONN, false>(s, s_act, s_chg, lx, ly, zz, nz, g.smask, theta, fresh, rej)) {
                    any = true;
                    chg |= 1u << zz;
                }
            }
            more = __syncthreads_or(any);
            it++;
        }
    } else {
        while (it < RELAX_ITCAP) {
            // pool the flagged voxels (18 / 26 neighbours per evaluation: lanes are worth keeping busy)
            const uint32_t a = col ? atomicExch(&s_act[ly][lx], 0u) : 0u;
            if (threadIdx.x == 0) s_qn[(it + 1) & 1] = 0; // the next sweep's counter: nobody touches it during this sweep
            if (a) {
                uint32_t off = atomicAdd(&s_qn[it & 1], (uint32_t)__popc(a));
                uint32_t m = a;
                while (m) {
                    const int zz = __ffs(m) - 1;
                    m &= m - 1;
                    s_queue[off++] = (uint16_t)((zz * TY + ly) * TX + lx);
                }
            }
            __syncthreads();
            const uint32_t T = s_qn[it & 1];
            if (T == 0) { more = false; break; } // uniform: nothing left to look at
            for (uint32_t i = threadIdx.x; i < T; i += 256) {
                const uint32_t code = s_queue[i];
                ws_eval<CONN, true>(s, s_act, s_chg, (int)(code % TX), (int)((code / TX) % TY), (int)(code / (TX * TY)), nz, g.smask, theta, fresh, rej);
            }
            __syncthreads();
            it++;
        }
    }
    static_assert(TX * TY == 256 && TX * TY * TZ <= 65536, "one lane per column; queue codes are 16 bits");
    if (more && threadIdx.x == 0) dirty[tile] = 1; // iteration cap: come back
    if (rej != NONE) atomicMin(&s_rej, rej);
    if (fresh) atomicAdd(&s_new, fresh);
    if (col) chg |= s_chg[ly][lx];
    // Which tiles read a changed voxel?  Inside the volume proper it is the lattice neighbour in the direction the voxel
    // leaves the box by: collect those directions in one 27-bit mask per workgroup and mark each tile once.  Only voxels
    // whose neighbour wraps around a row / slice end (scipy's linear-index neighbourhood) look their reader up one by one.
    uint32_t dirs = 0;
    for (int zz = 0; zz < nz && chg; zz++) {
        if (!((chg >> zz) & 1u)) continue;
        const int z = z0 + zz, y = y0 + ly, x = x0 + lx;
        C[(int64_t)z * g.hw + (int64_t)y * g.w + x] = (uint16_t)(s[((zz + 1) * BY + (ly + 1)) * BX + (lx + 1)] >> 16);
        const bool edge = lx == 0 || lx == TX - 1 || ly == 0 || ly == TY - 1 || zz == 0 || zz == TZ - 1 || x == (int)g.w - 1 ||
                          y == (int)g.h - 1 || z == (int)g.d - 1;
        if (!edge) continue; // every neighbour is inside this tile
#pragma unroll
        for (int k = 0; k < 27; k++) {
            if (!has_off<CONN>(g.smask, k)) continue;
            const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
            const int Z = z + dz, Y = y + dy, X = x + dx;
            if ((unsigned)X < (unsigned)g.w && (unsigned)Y < (unsigned)g.h) {
                if ((unsigned)Z >= (unsigned)g.d) continue; // outside the volume
                const int ex = X < x0 ? 0 : (X >= x0 + TX ? 2 : 1), ey = Y < y0 ? 0 : (Y >= y0 + TY ? 2 : 1),
                          ez = Z < z0 ? 0 : (Z >= z0 + TZ ? 2 : 1);
                dirs |= 1u << (ez * 9 + ey * 3 + ex); // bit 13 = this tile itself: ignored below
            } else { // wraps to the neighbouring row / slice
                const int64_t t = owner_tile(g, Z, Y, X);
                if (t >= 0) dirty[t] = 1;
            }
        }
    }
    if (dirs & ~(1u << 13)) atomicOr(&s_ev2, dirs);
    __syncthreads();
    if (threadIdx.x < 27 && threadIdx.x != 13 && ((s_ev2 >> threadIdx.x) & 1u)) {
        const int k = threadIdx.x;
        const int tz = z0 / TZ + k / 9 - 1, ty = y0 / TY + (k / 3) % 3 - 1, tx = x0 / TX + k % 3 - 1;
        if (tz >= 0 && tz < g.ntz && ty >= 0 && ty < g.nty && tx >= 0 && tx < g.ntx) dirty[((int64_t)tz * g.nty + ty) * g.ntx + tx] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_rej != NONE) {
            pending[tile] = 1;
            atomicMin(&st->minrej, s_rej);
        }
        // hot single-address atomics from a million visits cost milliseconds: only when somebody reads them
        if (s_new && theta < CINF) atomicAdd(&st->assigned, s_new); // the gate's "bulk is in" test
        if (theta_flags & 0x80000000u) atomicAdd(&st->sweeps, (uint32_t)it); // statistics (IVX_WS_TRACE)
    }
}

// the gate moved up: every parked tile is dirty again
__global__ __launch_bounds__(256) void k_ws_wake(int64_t ntiles, uint8_t *__restrict__ dirty, uint8_t *__restrict__ pending, WsState *st) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) st->minrej = NONE;
    if (t < ntiles && pending[t]) {
        pending[t] = 0;
        dirty[t] = 1;
    }
}

// entry flags: comp = ENTRY for entries (markers included), own index for the others
// also, per voxel, which neighbours can be its parents (lower cost, arc == its cost) and which can be zone members it
// touches (equal cost, arc <= cost): the per-level kernels then follow set bits instead of probing every neighbour
template <int CONN, typename MT>
__global__ __launch_bounds__(256) void k_ws_entries(WsGeom g, const uint16_t *__restrict__ I, const uint16_t *__restrict__ C,
                                                    const MT *__restrict__ mk, uint32_t *__restrict__ comp,
                                                    uint32_t *__restrict__ pmask, uint32_t *__restrict__ zmask) {
    __shared__ uint32_t s[NCELL];
    int z0, y0, x0;
    tile_origin(g, blockIdx.x, z0, y0, x0);
    load_tile(g, z0, y0, x0, I, C, s);
    __syncthreads();
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    if (!(x0 + lx < g.w && y0 + ly < g.h)) return;
    const int nz = min(TZ, (int)(g.d - z0));
    for (int zz = 0; zz < nz; zz++) {
        const int ci = ((zz + 1) * BY + (ly + 1)) * BX + (lx + 1);
        const uint32_t cell = s[ci];
        const uint32_t c = cell >> 16, iv = cell & 0xFFFFu;
        const int64_t p = (int64_t)(z0 + zz) * g.hw + (int64_t)(y0 + ly) * g.w + (x0 + lx);
        const bool marker = mk[p] != 0;
        uint32_t pm = 0, zm = 0;
#pragma unroll
        for (int k = 0; k < 27; k++) {
            if (!has_off<CONN>(g.smask, k)) continue;
            const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
            const uint32_t qv = s[ci + (dz * BY + dy) * BX + dx];
            const uint32_t qc = qv >> 16, w = absdiff(qv & 0xFFFFu, iv);
            // (a staged cell outside the volume carries cost CINF and intensity 0: it can be neither)
            const int64_t q = p + dz * g.hw + dy * g.w + dx;
            const bool in = q >= 0 && q < g.n;
            pm |= (in && qc < c && w == c) ? 1u << k : 0u;
            zm |= (in && qc == c && w <= c) ? 1u << k : 0u;
        }
        const bool e = marker || pm != 0;
        comp[p] = e ? ENTRY : (uint32_t)p;
        pmask[p] = marker ? 0u : pm;
        zmask[p] = zm;
    }
}

// x-runs inside a wave's 64 voxels: parent = start of the run (saves most of the unions on plateaus)
__global__ __launch_bounds__(256) void k_ws_runs(WsGeom g, const uint32_t *__restrict__ zmask, uint32_t *comp, int has_x) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool mine = p < g.n && comp[p] != ENTRY;
    bool link = false;
    // comp[p-1] is ENTRY, (p-1), or a value this kernel just wrote there: only its ENTRY-ness is read
    if (mine && has_x && lane > 0) link = ((zmask[p] >> 12) & 1u) && comp[p - 1] != ENTRY;
    const unsigned long long starts = __ballot(mine && !link);
    if (mine && link) {
        const unsigned long long below = starts & ((2ull << lane) - 1ull);
        const int s0 = 63 - __clzll(below);
        comp[p] = (uint32_t)(p - (lane - s0));
    }
}

__device__ __forceinline__ uint32_t ws_find(const uint32_t *comp, uint32_t a) {
    uint32_t r = __hip_atomic_load(&comp[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (r != a) {
        a = r;
        r = __hip_atomic_load(&comp[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return a;
}

__device__ __forceinline__ void ws_unite(uint32_t *comp, uint32_t a, uint32_t b) {
    for (;;) {
        a = ws_find(comp, a);
        b = ws_find(comp, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(&comp[b], a);
        if (old == b) return;
        b = old;
    }
}

template <int CONN>
__global__ __launch_bounds__(256) void k_ws_union(WsGeom g, const uint32_t *__restrict__ zmask, uint32_t *comp) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= g.n) return;
    if (__hip_atomic_load(&comp[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ENTRY) return;
    uint32_t zm = zmask[p] >> 14; // forward neighbours only (k = 14 .. 26)
    if ((threadIdx.x & 63) != 63) zm &= ~1u; // +x inside a wave: done by k_ws_runs
    while (zm) {
        const int k = 14 + __ffs(zm) - 1;
        zm &= zm - 1;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const int64_t q = p + dz * g.hw + dy * g.w + dx;
        if (__hip_atomic_load(&comp[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ENTRY) continue;
        ws_unite(comp, (uint32_t)p, (uint32_t)q);
    }
}

__global__ __launch_bounds__(256) void k_ws_flatten(int64_t n, uint32_t *comp) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t c = comp[p];
    if (c == ENTRY || c == (uint32_t)p) return;
    comp[p] = ws_find(comp, c);
}

// entries (markers excluded) per level: histogram, then scatter into the level's segment of elist.  A workgroup owns
// 16384 consecutive voxels and counts in LDS (levels below BK_LB; the rare higher ones go straight to global memory), so
// the hot global counters see one atomic per workgroup and level instead of one per wave and level.
constexpr int BK_LB = 4096, BK_CH = 64;
template <typename MT, bool SCATTER>
__global__ __launch_bounds__(256) void k_ws_bucket(int64_t n, const uint16_t *__restrict__ C, const MT *__restrict__ mk,
                                                   const uint32_t *__restrict__ comp, uint32_t *__restrict__ hist_or_cursor,
                                                   uint32_t *__restrict__ elist) {
    __shared__ uint32_t sh[BK_LB];
    for (int i = threadIdx.x; i < BK_LB; i += 256) sh[i] = 0;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * (256 * BK_CH);
    const int lane = threadIdx.x & 63;
    for (int pass = 0; pass < (SCATTER ? 2 : 1); pass++) {
        for (int j = 0; j < BK_CH; j++) {
            const int64_t p = b0 + (int64_t)j * 256 + threadIdx.x;
            const bool e = p < n && comp[p] == ENTRY && mk[p] == 0;
            const uint32_t c = e ? C[p] : 0;
            unsigned long long act = __ballot(e);
            while (act) {
                const int leader = __ffsll((long long)act) - 1;
                const uint32_t lc = __shfl(c, leader, 64);
                const unsigned long long same = __ballot(e && c == lc);
                const uint32_t cnt = (uint32_t)__popcll(same);
                uint32_t off = 0;
                if (lane == leader) {
                    if (lc < BK_LB) off = atomicAdd(&sh[lc], cnt);
                    else if (!SCATTER || pass == 1) off = atomicAdd(&hist_or_cursor[lc], cnt);
                }
                if (SCATTER && pass == 1) {
                    off = __shfl(off, leader, 64);
                    if (e && c == lc) elist[off + __popcll(same & ((1ull << lane) - 1ull))] = (uint32_t)p;
                }
                act &= ~same;
            }
        }
        __syncthreads();
        if (pass == 0) {
            // flush the counts (histogram) / turn them into this workgroup's base offsets (scatter)
            for (int i = threadIdx.x; i < BK_LB; i += 256) {
                const uint32_t v = sh[i];
                if (v) {
                    const uint32_t base = atomicAdd(&hist_or_cursor[i], v);
                    if (SCATTER) sh[i] = base;
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ uint32_t ws_tau_of(const uint32_t *__restrict__ comp, const uint32_t *tau, int64_t v) {
    const uint32_t cv = comp[v];
    return __hip_atomic_load(&tau[cv == ENTRY ? (uint32_t)v : cv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// key of every entry of level c: the earliest time stamp among its admissible parents (the set bits of pmask); mark the
// key as used
template <int CONN>
__global__ __launch_bounds__(256) void k_ws_keys(WsGeom g, const uint32_t *__restrict__ pmask, const uint32_t *__restrict__ comp,
                                                 const uint32_t *tau, const uint32_t *__restrict__ elist, uint32_t *__restrict__ key,
                                                 uint32_t *used, uint32_t start, uint32_t count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < count;
    uint32_t K = NONE;
    if (act) {
        const int64_t p = elist[start + i];
        uint32_t pm = pmask[p];
        while (pm) {
            const int k = __ffs(pm) - 1;
            pm &= pm - 1;
            const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
            K = min(K, ws_tau_of(comp, tau, p + dz * g.hw + dy * g.w + dx));
        }
        key[start + i] = K;
    }
    // one atomic per distinct key of the wave, and none for keys already marked
    unsigned long long todo = __ballot(act && K != NONE);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lk = __shfl(K, leader, 64);
        const unsigned long long same = __ballot(act && K == lk);
        if (lane == leader) {
            const uint32_t bit = 1u << (lk & 31);
            if (!(__hip_atomic_load(&used[lk >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&used[lk >> 5], bit);
        }
        todo &= ~same;
    }
}

// ONE workgroup: the used keys in ascending order, split into classes where the label changes; class j (ascending) gets
// time stamp base + (T-1-j): the larger the key, the earlier the pop.  Clears the bitmap for the next level.
// Every lane summarises its chunk of bitmap words as (first label, last label, label changes inside); summaries
// combine associatively, so the class index at the start of a chunk is an exclusive scan (wave shuffles + 16 wave totals).
struct WsSeg {
    int32_t first, last; // NOLAB = empty
    uint32_t flags;
};
__device__ __forceinline__ WsSeg ws_seg_join(const WsSeg &a, const WsSeg &b) {
    if (a.first == NOLAB) return b;
    if (b.first == NOLAB) return a;
    WsSeg r;
    r.first = a.first;
    r.last = b.last;
    r.flags = a.flags + b.flags + (a.last != b.first ? 1u : 0u);
    return r;
}
__global__ __launch_bounds__(1024) void k_ws_rank(WsState *st, uint32_t *used, uint32_t *__restrict__ remap, int32_t *lab,
                                                  uint32_t cap) {
    __shared__ WsSeg s_wave[16];
    __shared__ uint32_t s_total;
    const uint32_t base = st->base;
    const uint32_t nw = (base + 31) >> 5;
    const uint32_t chunk = (nw + 1023) / 1024;
    const uint32_t t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const uint32_t w0 = min(nw, t * chunk), w1 = min(nw, w0 + chunk);
    WsSeg mine = {NOLAB, NOLAB, 0};
    for (uint32_t w = w0; w < w1; w++) {
        uint32_t bits = used[w];
        while (bits) {
            const uint32_t k = w * 32 + (__ffs(bits) - 1);
            bits &= bits - 1;
            const int32_t l = lab[k];
            if (mine.first == NOLAB) mine.first = l;
            else if (l != mine.last) mine.flags++;
            mine.last = l;
        }
    }
    WsSeg inc = mine; // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        WsSeg up;
        up.first = __shfl_up(inc.first, o, 64);
        up.last = __shfl_up(inc.last, o, 64);
        up.flags = __shfl_up(inc.flags, o, 64);
        if (lane >= o) inc = ws_seg_join(up, inc);
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (t == 0) {
        WsSeg run = {NOLAB, NOLAB, 0};
        for (int i = 0; i < 16; i++) {
            const WsSeg w = s_wave[i];
            s_wave[i] = run; // exclusive prefix of wave i
            run = ws_seg_join(run, w);
        }
        s_total = run.first == NOLAB ? 0u : run.flags + 1u;
    }
    __syncthreads();
    const uint32_t T = s_total;
    if (T == 0) return;
    if ((uint64_t)base + T > cap) {
        if (t == 0) st->overflow = 1;
        return;
    }
    WsSeg ex; // everything before this lane's chunk
    ex.first = __shfl_up(inc.first, 1, 64);
    ex.last = __shfl_up(inc.last, 1, 64);
    ex.flags = __shfl_up(inc.flags, 1, 64);
    if (lane == 0) ex = WsSeg{NOLAB, NOLAB, 0};
    ex = ws_seg_join(s_wave[wv], ex);
    uint32_t cls = ex.flags;
    int32_t last = ex.last;
    for (uint32_t w = w0; w < w1; w++) {
        uint32_t bits = used[w];
        if (bits) used[w] = 0;
        while (bits) {
            const uint32_t k = w * 32 + (__ffs(bits) - 1);
            bits &= bits - 1;
            const int32_t l = lab[k];
            if (last != NOLAB && l != last) cls++;
            last = l;
            const uint32_t tn = base + (T - 1 - cls);
            remap[k] = tn;
            lab[tn] = l;
        }
    }
    __syncthreads();
    if (t == 0) st->base = base + T;
}

// entries of level c take their time stamp and hand it to the zones they touch (the set bits of zmask that are not
// entries themselves): the earliest stamp wins the zone
template <int CONN>
__global__ __launch_bounds__(256) void k_ws_claim(WsGeom g, const uint32_t *__restrict__ zmask, const uint32_t *__restrict__ comp,
                                                  uint32_t *tau, const uint32_t *__restrict__ elist, const uint32_t *__restrict__ key,
                                                  const uint32_t *__restrict__ remap, uint32_t start, uint32_t count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int64_t p = elist[start + i];
    const uint32_t K = key[start + i];
    if (K == NONE) return;
    const uint32_t t = remap[K];
    tau[p] = t;
    uint32_t zm = zmask[p];
    while (zm) {
        const int k = __ffs(zm) - 1;
        zm &= zm - 1;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const uint32_t r = comp[p + dz * g.hw + dy * g.w + dx];
        if (r == ENTRY) continue;
        if (__hip_atomic_load(&tau[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > t) atomicMin(&tau[r], t);
    }
}

template <typename MT>
__global__ __launch_bounds__(256) void k_ws_labels(int64_t n, const uint32_t *__restrict__ comp, const uint32_t *__restrict__ tau,
                                                   const int32_t *__restrict__ lab, MT *__restrict__ out, uint8_t *__restrict__ out8) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t cv = comp[p];
    const uint32_t t = tau[cv == ENTRY ? (uint32_t)p : cv];
    const int32_t l = t == NONE ? 0 : lab[t];
    if (out) out[p] = (MT)l;
    if (out8) out8[p] = (uint8_t)l;
}

__global__ void k_ws_fill32(uint32_t *p, int64_t n, uint32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_ws_set_hist0(uint32_t *hist, uint32_t m) { hist[0] = m; }
__global__ void k_ws_set_base(WsState *st, uint32_t m) { st->base = m; }

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct WsBufs {
    uint16_t *C;
    uint32_t *comp, *tau, *elist, *key, *pmask, *zmask, *hist, *cursor, *bcount, *bsum, *list, *used, *remap;
    int32_t *lab;
    uint8_t *dirty, *pending;
    WsState *st;
    uint32_t *total;
    size_t bytes;
};

static void ws_layout(const WsGeom &g, uint32_t cap, char *base, WsBufs *b) {
    size_t o = 0;
    auto take = [&](size_t n) { char *p = base ? base + o : nullptr; o += al(n); return p; };
    const int64_t nblk = cdiv(g.n, 2048);
    b->C = (uint16_t *)take((size_t)g.n * 2);
    b->comp = (uint32_t *)take((size_t)g.n * 4);
    b->tau = (uint32_t *)take((size_t)g.n * 4);
    b->elist = (uint32_t *)take((size_t)g.n * 4);
    b->key = (uint32_t *)take((size_t)g.n * 4);
    b->pmask = (uint32_t *)take((size_t)g.n * 4);
    b->zmask = (uint32_t *)take((size_t)g.n * 4);
    b->hist = (uint32_t *)take(65536 * 4);
    b->cursor = (uint32_t *)take(65536 * 4);
    b->bcount = (uint32_t *)take((size_t)(nblk + 1) * 4);
    b->bsum = (uint32_t *)take((size_t)(std::max<int64_t>(cdiv(nblk, 4096), 16) + 2) * 4);
    b->list = (uint32_t *)take((size_t)g.ntiles * 4);
    b->dirty = (uint8_t *)take((size_t)g.ntiles);
    b->pending = (uint8_t *)take((size_t)g.ntiles);
    b->used = (uint32_t *)take(((size_t)cap / 32 + 2) * 4);
    b->remap = (uint32_t *)take((size_t)cap * 4);
    b->lab = (int32_t *)take((size_t)cap * 4);
    b->st = (WsState *)take(sizeof(WsState));
    b->total = (uint32_t *)take(256);
    b->bytes = o;
}

static int make_geom(int64_t dz, int64_t dy, int64_t dx, const uint8_t *strct, WsGeom *g) {
    IVX_REQUIRE(dz > 0 && dy > 0 && dx > 0, IVX_EINVAL, "watershed_ift: empty volume");
    IVX_REQUIRE((double)dz * (double)dy * (double)dx < 4294967000.0, IVX_EINVAL, "watershed_ift: more than 2^32 voxels");
    g->d = dz; g->h = dy; g->w = dx; g->hw = dy * dx; g->n = dz * dy * dx;
    g->ntx = (int)cdiv(dx, TX); g->nty = (int)cdiv(dy, TY); g->ntz = (int)cdiv(dz, TZ);
    g->ntiles = (int64_t)g->ntx * g->nty * g->ntz;
    uint32_t m = 0;
    for (int k = 0; k < 27; k++)
        if (strct[k] && k != 13) m |= 1u << k;
    for (int k = 0; k < 27; k++) // the zone formulation needs an undirected neighbourhood
        IVX_REQUIRE(((m >> k) & 1u) == ((m >> (26 - k)) & 1u), IVX_EINVAL, "watershed_ift: structuring element must be symmetric");
    g->smask = m;
    return IVX_OK;
}

static int conn_of(uint32_t m) {
    uint32_t m6 = 0, m18 = 0;
    for (int k = 0; k < 27; k++) {
        if (k == 13) continue;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const int q = (dz != 0) + (dy != 0) + (dx != 0);
        if (q == 1) m6 |= 1u << k;
        if (q <= 2) m18 |= 1u << k;
    }
    if (m == m6) return 6;
    if (m == m18) return 18;
    if (m == (0x7FFFFFFu & ~(1u << 13))) return 26;
    return 0;
}

#define WS_CONN_SWITCH(conn, ...)  \
    switch (conn) {                 \
    case 6: { constexpr int CC = 6; __VA_ARGS__; } break;   \
    case 18: { constexpr int CC = 18; __VA_ARGS__; } break; \
    case 26: { constexpr int CC = 26; __VA_ARGS__; } break; \
    default: { constexpr int CC = 0; __VA_ARGS__; } break;  \
    }

struct WsTimer { // stage boundaries on the stream, read back once at the end (only when the caller asks for stats)
    hipEvent_t ev[8];
    int n = 0;
    bool on = false;
    void mark(hipStream_t st) {
        if (!on || n >= 8) return;
        if (hipEventCreate(&ev[n]) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(ev[n++], st);
    }
    void read(int64_t *out_us) {
        for (int i = 0; i + 1 < n; i++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            out_us[i] = (int64_t)(ms * 1000.0f);
        }
        for (int i = 0; i < n; i++) (void)hipEventDestroy(ev[i]);
        n = 0;
    }
};

template <typename MT>
static int ws_run(const WsGeom &g, const uint16_t *I, const MT *mk, MT *out, uint8_t *out8, uint16_t *cost_out, int64_t *stats,
                  hipStream_t st) {
    const int conn = conn_of(g.smask);
    const int64_t nblk = cdiv(g.n, 2048);
    const int gl = (int)cdiv(g.n, 256);
    // pass 0 needs the marker count before the tables can be sized: small fixed part first
    WsBufs b;
    uint32_t cap_guess = (uint32_t)std::min<int64_t>(g.n, (int64_t)1 << 22) + (1u << 22);
    ws_layout(g, cap_guess, nullptr, &b);
    void *mem = nullptr;
    IVX_REQUIRE(ws_get_s(WS_WSIFT, st, b.bytes, &mem) == IVX_OK, IVX_ENOMEM, "watershed_ift: %zu bytes of scratch", b.bytes);
    ws_layout(g, cap_guess, (char *)mem, &b);
    uint32_t cap = cap_guess;

    IVX_HIP(hipMemsetAsync(b.st, 0, sizeof(WsState), st));
    IVX_HIP(hipMemsetAsync(b.dirty, 0, (size_t)g.ntiles, st));
    IVX_HIP(hipMemsetAsync(b.hist, 0, 65536 * 4, st));
    hipLaunchKernelGGL(k_ws_init<MT>, dim3((unsigned)nblk), dim3(256), 0, st, g, mk, b.C, b.dirty, b.bcount, b.st);
    IVX_LAUNCH_CHECK();
    {
        const int rc = scan_u32_exclusive(b.bcount, nblk, b.bsum, b.total, st);
        if (rc != IVX_OK) return rc;
    }
    uint32_t M = 0;
    WsState hs;
    IVX_HIP(hipMemcpyAsync(&M, b.total, 4, hipMemcpyDeviceToHost, st));
    IVX_HIP(hipMemcpyAsync(&hs, b.st, sizeof(hs), hipMemcpyDeviceToHost, st));
    IVX_HIP(hipStreamSynchronize(st));
    IVX_REQUIRE(!hs.neg, IVX_EINVAL, "watershed_ift: negative markers are not supported (the reference passes 0 / 1 / 2)");
    if (M == 0) { // nothing to flood: labels = markers = 0
        if (out) IVX_HIP(hipMemsetAsync(out, 0, (size_t)g.n * sizeof(MT), st));
        if (out8) IVX_HIP(hipMemsetAsync(out8, 0, (size_t)g.n, st));
        if (cost_out) IVX_HIP(hipMemsetAsync(cost_out, 0xFF, (size_t)g.n * 2, st));
        if (stats) memset(stats, 0, 16 * sizeof(int64_t));
        return IVX_OK;
    }
    if ((uint64_t)M + (1u << 22) > cap) { // many marker voxels: larger time-stamp tables (layout changes only behind `used`)
        cap = (uint32_t)std::min<uint64_t>((uint64_t)M + (1u << 22), 0xFFFFFFF0ull);
        ws_layout(g, cap, nullptr, &b);
        IVX_REQUIRE(ws_get_s(WS_WSIFT, st, b.bytes, &mem) == IVX_OK, IVX_ENOMEM, "watershed_ift: %zu bytes of scratch", b.bytes);
        ws_layout(g, cap, (char *)mem, &b);
        // the grow may have moved the block: start over (rare path)
        IVX_HIP(hipMemsetAsync(b.st, 0, sizeof(WsState), st));
        IVX_HIP(hipMemsetAsync(b.dirty, 0, (size_t)g.ntiles, st));
        IVX_HIP(hipMemsetAsync(b.hist, 0, 65536 * 4, st));
        hipLaunchKernelGGL(k_ws_init<MT>, dim3((unsigned)nblk), dim3(256), 0, st, g, mk, b.C, b.dirty, b.bcount, b.st);
        IVX_LAUNCH_CHECK();
        const int rc = scan_u32_exclusive(b.bcount, nblk, b.bsum, b.total, st);
        if (rc != IVX_OK) return rc;
    }
    IVX_HIP(hipMemsetAsync(b.used, 0, ((size_t)cap / 32 + 2) * 4, st));

    WsTimer tm;
    tm.on = stats != nullptr;
    tm.mark(st);
    // ---- 1. costs ------------------------------------------------------------------------------------------
    int64_t rounds = 0, visits = 0;
    IVX_HIP(hipMemsetAsync(b.pending, 0, (size_t)g.ntiles, st));
    hipLaunchKernelGGL(k_ws_wake, dim3(1), dim3(256), 0, st, (int64_t)0, b.dirty, b.pending, b.st); // minrej = NONE
    IVX_LAUNCH_CHECK();
    const char *gate_env = getenv("IVX_WS_GATE");
    const bool gate = gate_env && gate_env[0] == '1'; // measured slower on the noise phantom (more rounds AND more visits): opt-in
    const bool trace = getenv("IVX_WS_TRACE") != nullptr;
    uint32_t theta = gate ? 0u : CINF;
    for (;;) {
        IVX_HIP(hipMemsetAsync(&b.st->nlist, 0, 4, st));
        hipLaunchKernelGGL(k_ws_build_list, dim3((unsigned)cdiv(g.ntiles, 256)), dim3(256), 0, st, g.ntiles, b.dirty, b.list, b.st);
        IVX_LAUNCH_CHECK();
        uint32_t seq = 0, msg[3] = {0, 0, 0};
        int rc = mailbox_publish(&b.st->nlist, 3, st, &seq);
        if (rc != IVX_OK) return rc;
        rc = mailbox_wait(seq, st, msg, 3);
        if (rc != IVX_OK) return rc;
        const uint32_t nl = msg[0];
        if (trace) fprintf(stderr, "ws round %lld theta %u tiles %u minrej %u assigned %u\n", (long long)rounds, theta, nl, msg[1], msg[2]);
        if (nl == 0) {
            if (theta >= CINF || msg[1] == NONE) break; // nothing was refused: this is the fix-point
            // converged below the gate: lift it to the first level that has work, or all the way once the bulk is in
            theta = (uint64_t)msg[2] * 2 > (uint64_t)g.n ? CINF : msg[1];
            hipLaunchKernelGGL(k_ws_wake, dim3((unsigned)cdiv(g.ntiles, 256)), dim3(256), 0, st, g.ntiles, b.dirty, b.pending, b.st);
            IVX_LAUNCH_CHECK();
            continue;
        }
        rounds++;
        visits += nl;
        WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_relax<CC>, dim3(nl), dim3(256), 0, st, g, I, b.C, b.list, b.dirty, b.pending, b.st, theta | (trace ? 0x80000000u : 0u)));
        IVX_LAUNCH_CHECK();
        IVX_REQUIRE(rounds < 1000000, IVX_EHIP, "watershed_ift: relaxation does not terminate");
    }
    if (cost_out) IVX_HIP(hipMemcpyAsync(cost_out, b.C, (size_t)g.n * 2, hipMemcpyDeviceToDevice, st));

    tm.mark(st);
    // ---- 2. entries and zones ------------------------------------------------------------------------------
    WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_ws_entries<CC, MT>), dim3((unsigned)g.ntiles), dim3(256), 0, st, g, I, b.C, mk, b.comp, b.pmask, b.zmask));
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ws_runs, dim3(gl), dim3(256), 0, st, g, b.zmask, b.comp, (int)((g.smask >> 12) & 1u));
    IVX_LAUNCH_CHECK();
    WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_union<CC>, dim3(gl), dim3(256), 0, st, g, b.zmask, b.comp));
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ws_flatten, dim3(gl), dim3(256), 0, st, g.n, b.comp);
    IVX_LAUNCH_CHECK();

    tm.mark(st);
    // ---- 3. entries by level -------------------------------------------------------------------------------
    hipLaunchKernelGGL((k_ws_bucket<MT, false>), dim3((unsigned)cdiv(g.n, 256 * BK_CH)), dim3(256), 0, st, g.n, b.C, mk, b.comp, b.hist, b.elist);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ws_set_hist0, dim3(1), dim3(1), 0, st, b.hist, M);
    std::vector<uint32_t> hist(65536);
    IVX_HIP(hipMemcpyAsync(hist.data(), b.hist, 65536 * 4, hipMemcpyDeviceToHost, st));
    IVX_HIP(hipMemcpyAsync(b.cursor, b.hist, 65536 * 4, hipMemcpyDeviceToDevice, st));
    {
        const int rc = scan_u32_exclusive(b.cursor, 65536, b.bsum, b.total, st);
        if (rc != IVX_OK) return rc;
    }
    hipLaunchKernelGGL((k_ws_marker_list<MT>), dim3((unsigned)nblk), dim3(256), 0, st, g, mk, b.bcount, b.elist, b.key, b.lab);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_ws_bucket<MT, true>), dim3((unsigned)cdiv(g.n, 256 * BK_CH)), dim3(256), 0, st, g.n, b.C, mk, b.comp, b.cursor, b.elist);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ws_fill32, dim3(2048), dim3(256), 0, st, b.tau, g.n, NONE);
    IVX_LAUNCH_CHECK();
    IVX_HIP(hipStreamSynchronize(st)); // hist is on the host now

    tm.mark(st);
    // ---- 4. the level chain --------------------------------------------------------------------------------
    hipLaunchKernelGGL(k_ws_fill_used, dim3((unsigned)cdiv((M + 31) / 32, 256)), dim3(256), 0, st, b.used, M);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ws_set_base, dim3(1), dim3(1), 0, st, b.st, M);
    IVX_LAUNCH_CHECK();
    int64_t nlevels = 0;
    uint32_t start = 0;
    for (uint32_t c = 0; c < 65536; c++) {
        const uint32_t cnt = hist[c];
        if (!cnt) continue;
        nlevels++;
        const unsigned gb = (unsigned)cdiv(cnt, 256);
        if (c > 0) {
            WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_keys<CC>, dim3(gb), dim3(256), 0, st, g, b.pmask, b.comp, b.tau, b.elist, b.key,
                                                      b.used, start, cnt));
            IVX_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_ws_rank, dim3(1), dim3(1024), 0, st, b.st, b.used, b.remap, b.lab, cap);
        IVX_LAUNCH_CHECK();
        WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_claim<CC>, dim3(gb), dim3(256), 0, st, g, b.zmask, b.comp, b.tau, b.elist, b.key,
                                                  b.remap, start, cnt));
        IVX_LAUNCH_CHECK();
        start += cnt;
    }

    tm.mark(st);
    // ---- 5. labels -----------------------------------------------------------------------------------------
    hipLaunchKernelGGL(k_ws_labels<MT>, dim3(gl), dim3(256), 0, st, g.n, b.comp, b.tau, b.lab, out, out8);
    IVX_LAUNCH_CHECK();
    tm.mark(st);
    IVX_HIP(hipMemcpyAsync(&hs, b.st, sizeof(hs), hipMemcpyDeviceToHost, st));
    IVX_HIP(hipStreamSynchronize(st));
    IVX_REQUIRE(!hs.overflow, IVX_ENOMEM, "watershed_ift: more than %u time-stamp classes", cap);
    if (stats) {
        stats[0] = rounds; stats[1] = visits; stats[2] = nlevels; stats[3] = hs.base; stats[4] = M; stats[5] = start;
        stats[6] = g.ntiles; stats[7] = hs.sweeps;
        for (int i = 8; i < 16; i++) stats[i] = 0;
        tm.read(stats + 8); // [8] costs, [9] zones, [10] bucketing, [11] level chain, [12] labels (microseconds)
    }
    return IVX_OK;
}

} // namespace

extern "C" int ivx_dev_watershed_ift(const uint16_t *cost, int mdtype, const void *markers, int64_t dz, int64_t dy, int64_t dx,
                                     const uint8_t strct[27], void *out_labels, uint8_t *out_u8, uint16_t *cost_out,
                                     int64_t stats[16], void *stream) {
    WsGeom g;
    const int rc = make_geom(dz, dy, dx, strct, &g);
    if (rc != IVX_OK) return rc;
    IVX_REQUIRE(mdtype == IVX_I16 || mdtype == IVX_I8, IVX_EINVAL, "watershed_ift: markers must be int16 or int8");
    IVX_REQUIRE(cost && markers && (out_labels || out_u8), IVX_EINVAL, "watershed_ift: null buffer");
    if (mdtype == IVX_I16) return ws_run<int16_t>(g, cost, (const int16_t *)markers, (int16_t *)out_labels, out_u8, cost_out, stats, S(stream));
    return ws_run<int8_t>(g, cost, (const int8_t *)markers, (int8_t *)out_labels, out_u8, cost_out, stats, S(stream));
}

__global__ void k_ws_widen(const uint8_t *__restrict__ in, uint16_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

extern "C" int ivx_watershed_ift(int idtype, const void *input, const int64_t shape[3], int mdtype, const void *markers,
                                 const uint8_t strct[27], void *output, uint16_t *cost_out, int64_t stats[16]) {
    HostCallGuard guard;
    IVX_REQUIRE(idtype == IVX_U8 || idtype == IVX_U16, IVX_EINVAL, "watershed_ift: input must be uint8 or uint16 (scipy raises TypeError)");
    IVX_REQUIRE(mdtype == IVX_I16 || mdtype == IVX_I8, IVX_EINVAL, "watershed_ift: markers must be int16 or int8");
    const int64_t n = shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    const size_t msz = mdtype == IVX_I16 ? 2 : 1;
    void *dI = nullptr, *dM = nullptr, *dO = nullptr, *dC = nullptr, *dT = nullptr;
    int rc;
    if ((rc = ws_get(WS_IN, (size_t)n * 2, &dI)) != IVX_OK) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)n * msz, &dM)) != IVX_OK) return rc;
    if ((rc = ws_get(WS_OUT, (size_t)n * msz, &dO)) != IVX_OK) return rc;
    if (cost_out && (rc = ws_get(WS_AUX1, (size_t)n * 2, &dC)) != IVX_OK) return rc;
    if (idtype == IVX_U8) {
        if ((rc = ws_get(WS_AUX2, (size_t)n, &dT)) != IVX_OK) return rc;
        IVX_HIP(hipMemcpy(dT, input, (size_t)n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_ws_widen, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, 0, (const uint8_t *)dT, (uint16_t *)dI, n);
        IVX_LAUNCH_CHECK();
    } else {
        IVX_HIP(hipMemcpy(dI, input, (size_t)n * 2, hipMemcpyHostToDevice));
    }
    IVX_HIP(hipMemcpy(dM, markers, (size_t)n * msz, hipMemcpyHostToDevice));
    rc = ivx_dev_watershed_ift((const uint16_t *)dI, mdtype, dM, shape[0], shape[1], shape[2], strct, dO, nullptr, (uint16_t *)dC,
                               stats, nullptr);
    if (rc != IVX_OK) return rc;
    IVX_HIP(hipMemcpy(output, dO, (size_t)n * msz, hipMemcpyDeviceToHost));
    if (cost_out) IVX_HIP(hipMemcpy(cost_out, dC, (size_t)n * 2, hipMemcpyDeviceToHost));
    return IVX_OK;
}
