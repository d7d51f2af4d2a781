// ws_tiles.h -- what the two marker floods share (k_wsift.hip: scipy's IFT flood, k_wssk.hip: scikit-image's heap flood):
// volume geometry, the 16x16x8 tile staged in LDS with its halo, the chaotic min-max relaxation of the path cost over
// dirty tiles, the per-level bucketing of voxels, small helpers.  Everything is templated on SK:
//   SK = false  scipy.ndimage.watershed_ift: arc cost |I(p) - I(q)|, markers cost 0, neighbours by LINEAR index
//               (row / slice wrap-around neighbours are real neighbours there);
//   SK = true   skimage.segmentation.watershed: a path costs the largest image value ON it, markers cost their own
//               value, neighbours are lattice neighbours inside the volume.
// Included by exactly those two translation units; all symbols have internal linkage.
#pragma once
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
    uint32_t imax;     // (scikit-image branch) largest image value: no cost is above it
    uint32_t nlist;    // dirty tiles of the next round            } read by the host
    uint32_t minrej;   // smallest cost refused by the gate so far } after every round
    uint32_t assigned; // voxels that have a finite cost           } (one mailbox message)
    uint32_t sweeps;   // LDS sweeps over all tile visits (statistics)
    uint32_t nlist_b;  // the list length of every other round (k_ws_build_list: a round counts into one and clears the other)
    uint32_t ticket;   // workgroups of k_ws_build_list that are done (the last one publishes)
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
template <bool SK> __device__ __forceinline__ bool cell_ok(const WsGeom &g, int64_t z, int64_t y, int64_t x, int64_t L) {
    if (SK) return (uint64_t)x < (uint64_t)g.w && (uint64_t)y < (uint64_t)g.h && (uint64_t)z < (uint64_t)g.d; // lattice neighbours only
    return L >= 0 && L < g.n;                                                                                   // scipy: by linear index
}

template <bool SK>
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
                const int64_t zz = z0 + lz - 1, yy = y0 + ly - 1, xx = x0 + half * 8;
                const int64_t L = zz * g.hw + yy * g.w + xx;
                uint32_t *dst = s + row * BX + 1 + half * 8;
                if (SK ? (cell_ok<true>(g, zz, yy, xx, L) && xx + 8 <= g.w) : (L >= 0 && L + 8 <= g.n)) {
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
                        dst[e] = cell_ok<SK>(g, zz, yy, xx + e, Le) ? ((uint32_t)C[Le] << 16) | I[Le] : CINF << 16;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < (NROW * 2 + 255) / 256; q++) { // the two x-halo cells of every row
            const int it = threadIdx.x + q * 256;
            if (it < NROW * 2) {
                const int row = it >> 1, side = it & 1, ly = row % BY, lz = row / BY;
                const int64_t zz = z0 + lz - 1, yy = y0 + ly - 1, xx = x0 + (side ? TX : -1);
                const int64_t L = zz * g.hw + yy * g.w + xx;
                s[row * BX + (side ? BX - 1 : 0)] = cell_ok<SK>(g, zz, yy, xx, L) ? ((uint32_t)C[L] << 16) | I[L] : CINF << 16;
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
        const bool ok = c < NCELL && cell_ok<SK>(g, z0 + lz - 1, y0 + ly - 1, x0 + lx - 1, L);
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

template <typename MT, bool SK>
__global__ __launch_bounds__(256) void k_ws_init(WsGeom g, const MT *__restrict__ mk, const uint16_t *__restrict__ I,
                                                 uint16_t *__restrict__ C, uint8_t *__restrict__ dirty,
                                                 uint32_t *__restrict__ bcount, WsState *st) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * 2048;
    uint32_t mine = 0, vmax = 0;
    bool neg = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int64_t p = b0 + j * 256 + threadIdx.x;
        if (p >= g.n) continue;
        const int m = (int)mk[p];
        C[p] = m ? (SK ? I[p] : (uint16_t)0) : (uint16_t)CINF; // a marker's cost is final from the start
        if (SK && I[p] == (uint16_t)CINF) st->overflow = 1;       // 65535 is the "never reached" cost
        if (SK) vmax = max(vmax, (uint32_t)I[p]);
        if (m) {
            mine++;
            neg |= m < 0;
            const int64_t z = p / g.hw, r = p - z * g.hw, y = r / g.w, x = r - y * g.w;
            dirty[((z / TZ) * g.nty + y / TY) * g.ntx + x / TX] = 1;
            // A marker never changes, so a tile that only sees it in its halo would not be woken by it: when the marker
            // sits on a face of its tile, wake the tiles that own its neighbours (normally the marker's in-tile
            // neighbours on that face change and do this; a one-voxel-wide volume has none).
            if (x % TX == 0 || x % TX == TX - 1 || y % TY == 0 || y % TY == TY - 1 || z % TZ == 0 || z % TZ == TZ - 1)
                for (int k = 0; k < 27; k++) {
                    if (k == 13 || !((g.smask >> k) & 1u)) continue;
                    const int64_t Z = z + k / 9 - 1, Y = y + (k / 3) % 3 - 1, X = x + k % 3 - 1;
                    if (SK && !((uint64_t)X < (uint64_t)g.w && (uint64_t)Y < (uint64_t)g.h && (uint64_t)Z < (uint64_t)g.d)) continue;
                    const int64_t t = owner_tile(g, Z, Y, X);
                    if (t >= 0) dirty[t] = 1;
                }
        }
    }
    if (mine) atomicAdd(&s_cnt, mine);
    if (neg) st->neg = 1;
    if (SK) { // (the bucket passes size their LDS counters by it)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, o, 64));
        if ((threadIdx.x & 63) == 0 && vmax > __hip_atomic_load(&st->imax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->imax, vmax);
    }
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

// The tiles that hold a voxel without a cost yet (CINF): behind the level floods -- whose costs are final -- the relaxation has
// nothing to do anywhere else, and "every tile looks once" was a pass of 3 240-cell stagings over a volume that is 95 % done
// (first round 0.83 ms at 512^3, 6.6 ms at 1024^3).  Lane = 8 voxels of a row (rows are whole 64s on this path).
__global__ __launch_bounds__(256) void k_ws_mark_open_tiles(WsGeom g, const uint16_t *__restrict__ C, uint8_t *__restrict__ dirty) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.n / 8) return;
    const uint4 v = reinterpret_cast<const uint4 *>(C)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    bool open = false;
#pragma unroll
    for (int k = 0; k < 4; k++) open |= (w[k] & 0xFFFFu) == CINF || (w[k] >> 16) == CINF;
    if (!open) return;
    const int64_t p = i * 8, z = p / g.hw, r = p - z * g.hw, y = r / g.w, x = r - y * g.w;
    dirty[((z / TZ) * g.nty + y / TY) * g.ntx + x / TX] = 1;
}

// dirty flags -> list.  Lane = 16 tiles (one 16-byte load), one atomic per workgroup: at 1024^3 (2^19 tiles) the lane-per-tile
// form with an atomic per wave took 26 us a round, 170 rounds a flood.
constexpr int BL_PER = 16;
// With a mailbox slot (round 6): the launch also does what a 4-byte fill in front of it and a publishing kernel behind it did --
// the round counts into `cur`, the last workgroup to finish (a ticket) clears `other` for the next round and writes
// { length, minrej, assigned } to the host's mailbox itself: one launch per round instead of three (~10 us, 40 - 180 rounds a flood).
__global__ __launch_bounds__(256) void k_ws_build_list(int64_t ntiles, uint8_t *__restrict__ dirty, uint32_t *__restrict__ list,
                                                       WsState *st, uint32_t *cur, uint32_t *other, uint32_t *mb, uint32_t seq) {
    __shared__ uint32_t s_wave[4], s_base;
    const int64_t t0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * BL_PER;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t m = 0; // which of my 16 tiles are dirty
    if (t0 + BL_PER <= ntiles) {
        const uint4 q = *reinterpret_cast<const uint4 *>(dirty + t0);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int b = 0; b < 4; b++) m |= ((w[k] >> (8 * b)) & 0xFFu ? 1u : 0u) << (4 * k + b);
        if (m) *reinterpret_cast<uint4 *>(dirty + t0) = make_uint4(0, 0, 0, 0);
    } else {
        for (int k = 0; k < BL_PER; k++)
            if (t0 + k < ntiles && dirty[t0 + k]) {
                m |= 1u << k;
                dirty[t0 + k] = 0;
            }
    }
    const uint32_t cnt = (uint32_t)__popc(m);
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = tot ? atomicAdd(cur, tot) : 0u;
    }
    __syncthreads();
    uint32_t off = s_base + inc - cnt;
    for (int q = 0; q < wv; q++) off += s_wave[q];
    while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1;
        list[off++] = (uint32_t)(t0 + k);
    }
    if (!mb) return;
    __threadfence(); // (this workgroup's count is in before its ticket)
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&st->ticket, 1u) == gridDim.x - 1) {
        st->ticket = 0;
        *other = 0;
        mb[0] = __hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mb[1] = __hip_atomic_load(&st->minrej, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mb[2] = __hip_atomic_load(&st->assigned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __hip_atomic_store(&mb[63], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// one round's list, counted into the counter of the round's parity and published by the launch itself
static int ws_build_list_publish(int64_t ntiles, uint8_t *dirty, uint32_t *list, WsState *wst, int parity, hipStream_t st, uint32_t *seq_out,
                                 uint32_t **cur_out) {
    uint32_t *slot = nullptr;
    const int rc = mailbox_reserve(&slot, seq_out);
    if (rc != IVX_OK) return rc;
    uint32_t *cur = parity ? &wst->nlist_b : &wst->nlist, *other = parity ? &wst->nlist : &wst->nlist_b;
    hipLaunchKernelGGL(k_ws_build_list, dim3((unsigned)cdiv(ntiles, 256 * BL_PER)), dim3(256), 0, st, ntiles, dirty, list, wst, cur, other, slot, *seq_out);
    IVX_LAUNCH_CHECK();
    *cur_out = cur;
    return IVX_OK;
}

// one visit of a dirty tile: relax to the local fix-point, write the changed costs back, wake the tiles that read them.
// The first sweep evaluates every voxel (lane = z-column); after that only voxels a neighbour of which changed are
// looked at again -- a flag word per z-column in LDS -- and those few are pooled into a work queue in LDS and dealt out
// one per lane (sparse sweeps walked column by column kept one or two lanes of a wave busy: 1.6e9 of the 3.7e9 voxel
// evaluations of a 512^3 flood, at a tenth of the lanes).
// theta gates the flood: a cost above it is not accepted yet (the tile is parked in `pending`), so that below the level
// where the bulk of the volume connects only final costs spread -- no wave of provisional costs to correct later.
template <int CONN, bool LDS_CHG, bool SK>
__device__ __forceinline__ bool ws_eval(uint32_t *s, uint32_t (*s_act)[TX], uint32_t (*s_chg)[TX], int lx, int ly, int zz, int nz,
                                        uint32_t smask, uint32_t theta, uint32_t &fresh, uint32_t &rej) {
    const int ci = ((zz + 1) * BY + (ly + 1)) * BX + (lx + 1);
    const uint32_t cell = s[ci];
    const uint32_t c = cell >> 16, iv = cell & 0xFFFFu;
    if (c == (SK ? iv : 0u)) return false; // at its floor already (markers always are)
    uint32_t best = c;
#pragma unroll
    for (int k = 0; k < 27; k++) {
        if (!has_off<CONN>(smask, k)) continue;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const uint32_t qv = s[ci + (dz * BY + dy) * BX + dx];
        const uint32_t m = max(qv >> 16, SK ? iv : absdiff(qv & 0xFFFFu, iv));
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

template <int CONN, bool SK>
__global__ __launch_bounds__(256) void k_ws_relax(WsGeom g, const uint16_t *__restrict__ I, uint16_t *C,
                                                  const uint32_t *__restrict__ list, uint8_t *dirty, uint8_t *pending,
                                                  WsState *st, uint32_t theta_flags, uint32_t offset, const uint32_t *nlist) {
    __shared__ uint32_t s[NCELL];
    __shared__ uint32_t s_act[TY][TX], s_chg[TY][TX];
    __shared__ uint16_t s_queue[CONN == 6 ? 2 : TX * TY * TZ]; // (the 6-neighbour form never pools)
    __shared__ uint32_t s_qn[2];
    __shared__ uint32_t s_new, s_rej, s_ev2;
    // (the grid may be a guess made before the host knew the list's length -- ws_cost_rounds: workgroups beyond the list leave)
    if (blockIdx.x + offset >= *nlist) return;
    const int64_t tile = list[blockIdx.x + offset];
    const uint32_t theta = theta_flags & 0xFFFFu; // bit 31 of the argument: collect the sweep statistic
    int z0, y0, x0;
    tile_origin(g, tile, z0, y0, x0);
    load_tile<SK>(g, z0, y0, x0, I, C, s);
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
            if (ws_eval<CONN, false, SK>(s, s_act, s_chg, lx, ly, zz, nz, g.smask, theta, fresh, rej)) chg |= 1u << zz;
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
                if (ws_eval<CONN, false, SK>(s, s_act, s_chg, lx, ly, zz, nz, g.smask, theta, fresh, rej)) {
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
                ws_eval<CONN, true, SK>(s, s_act, s_chg, (int)(code % TX), (int)((code / TX) % TY), (int)(code / (TX * TY)), nz, g.smask, theta, fresh, rej);
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
            } else if (!SK) { // wraps to the neighbouring row / slice (scipy's linear-index neighbourhood only)
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
// entries (markers excluded) per level: histogram, then scatter into the level's segment of elist.  A workgroup owns
// 16384 consecutive voxels and counts in LDS (levels below BK_LB; the rare higher ones go straight to global memory), so
// the hot global counters see one atomic per workgroup and level instead of one per wave and level.
constexpr int BK_LB = 4096, BK_CH = 64;
// PRED: device functor, pred(p) = "voxel p goes into its level's list"
template <typename PRED, bool SCATTER>
__global__ __launch_bounds__(256) void k_ws_bucket(int64_t n, const uint16_t *__restrict__ C, PRED pred,
                                                   uint32_t *__restrict__ hist_or_cursor, uint32_t *__restrict__ elist, int lb) {
    // lb counters in LDS (dynamic, lb * 4 bytes: BK_LB at most -- a caller that knows its highest level asks for fewer: 16 KB of LDS
    // a workgroup and 4096 counters to clear and flush per 16 384 voxels are most of the pass when the levels end at 255)
    extern __shared__ uint32_t sh[];
    for (int i = threadIdx.x; i < lb; i += 256) sh[i] = 0;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * (256 * BK_CH);
    // One LDS atomic per voxel that goes into a list: the hardware serialises lanes that hit the same counter inside the
    // instruction, and its return value IS the voxel's slot.  (Before: a match-any loop per wave -- shuffle, ballot, one LDS
    // atomic by a leader lane per distinct level among the wave's voxels, ~20 - 30 dependent iterations on a noise volume --
    // which made these streaming passes 5 - 10x slower than their bytes: 2.2 + 0.8 ms at 512^3 for 0.9 GB.)
    // (Round 6: a wave whose voxels all sit on ONE level -- the inside of a plateau, most of a windowed gradient -- takes one atomic
    // for the wave instead of 64 serialised ones on the same counter: the passes over a 1024^3 volume were 3 - 6 ms each for 3 GB.)
    const int lane = threadIdx.x & 63;
    for (int pass = 0; pass < (SCATTER ? 2 : 1); pass++) {
        for (int j = 0; j < BK_CH; j++) {
            const int64_t p = b0 + (int64_t)j * 256 + threadIdx.x;
            const bool e = p < n && pred(p);
            const uint32_t c = e ? (uint32_t)C[p] : 0xFFFFFFFFu;
            const unsigned long long act = __ballot(e);
            if (!act) continue;
            const int lead = __builtin_ctzll(act);
            const uint32_t c0 = (uint32_t)__shfl((int)c, lead, 64);
            uint32_t off = 0;
            bool have = false;
            if (__ballot(c == c0) == act) {
                uint32_t base = 0;
                if (c0 < (uint32_t)lb) {
                    if (lane == lead) base = atomicAdd(&sh[c0], (uint32_t)__popcll(act));
                    have = e;
                } else if (!SCATTER || pass == 1) {
                    if (lane == lead) base = atomicAdd(&hist_or_cursor[c0], (uint32_t)__popcll(act));
                    have = e;
                }
                off = (uint32_t)__shfl((int)base, lead, 64) + (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
            } else if (e) {
                if (c < (uint32_t)lb) {
                    off = atomicAdd(&sh[c], 1u);
                    have = true;
                } else if (!SCATTER || pass == 1) {
                    off = atomicAdd(&hist_or_cursor[c], 1u);
                    have = true;
                }
            }
            if (have && SCATTER && pass == 1) elist[off] = (uint32_t)p;
        }
        __syncthreads();
        if (pass == 0) {
            // flush the counts (histogram) / turn them into this workgroup's base offsets (scatter)
            for (int i = threadIdx.x; i < lb; i += 256) {
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
__global__ void k_ws_fill32(uint32_t *p, int64_t n, uint32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

static int make_geom(int64_t dz, int64_t dy, int64_t dx, const uint8_t *strct, WsGeom *g) {
    IVX_REQUIRE(dz > 0 && dy > 0 && dx > 0, IVX_EINVAL, "watershed: empty volume");
    IVX_REQUIRE((double)dz * (double)dy * (double)dx < 4294967000.0, IVX_EINVAL, "watershed: more than 2^32 voxels");
    g->d = dz; g->h = dy; g->w = dx; g->hw = dy * dx; g->n = dz * dy * dx;
    g->ntx = (int)cdiv(dx, TX); g->nty = (int)cdiv(dy, TY); g->ntz = (int)cdiv(dz, TZ);
    g->ntiles = (int64_t)g->ntx * g->nty * g->ntz;
    uint32_t m = 0;
    for (int k = 0; k < 27; k++)
        if (strct[k] && k != 13) m |= 1u << k;
    for (int k = 0; k < 27; k++) // the zone formulation needs an undirected neighbourhood
        IVX_REQUIRE(((m >> k) & 1u) == ((m >> (26 - k)) & 1u), IVX_EINVAL, "watershed: structuring element must be symmetric");
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

// ---- union-find over voxels: comp[p] = ENTRY (not a member) or a parent index; zmask[p] = the neighbours (bit k =
// (dz+1)*9 + (dy+1)*3 + (dx+1)) that belong to the same set ----
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
    const int lane = threadIdx.x & 63;
    // Whether a voxel is a member at all (comp != ENTRY) was settled by the launches before this one and never changes here: those
    // tests are ordinary loads that the L2 may serve; only the parent pointers that other workgroups rewrite meanwhile are read at
    // agent scope (ws_find).
    const uint32_t *member = comp;
    const bool mine = p < g.n && member[p] != ENTRY;
    const uint32_t zmp = mine ? zmask[p] : 0u;
    // A +y / +z link p -> q is implied by its left neighbour's when p-1 ~ p and q-1 ~ q are links themselves and p-1 -> q-1 is the
    // same kind of link: the sets are the same with or without it (round 5: inside a plateau only the first voxel of a row segment
    // hooks across, the others skip two finds and an atomic each).  The left neighbour's mask comes by shuffle, so lane 0 never skips.
    const uint32_t zleft = __shfl_up(zmp, 1, 64);
    const bool left_ok = lane > 0 && (zmp >> 12 & 1u) && __shfl_up((int)mine, 1, 64);
    if (!mine) return;
    uint32_t zm = zmp >> 14; // forward neighbours only (k = 14 .. 26)
    if (lane != 63) zm &= ~1u; // +x inside a wave: done by k_ws_runs
    while (zm) {
        const int k = 14 + __ffs(zm) - 1;
        zm &= zm - 1;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const int64_t q = p + dz * g.hw + dy * g.w + dx;
        if (member[q] == ENTRY) continue;
        if ((k == 16 || k == 22) && left_ok && (zleft >> k & 1u) && (zmask[q] >> 12 & 1u) && member[q - 1] != ENTRY) continue;
        ws_unite(comp, (uint32_t)p, (uint32_t)q);
    }
}

// ---- the same union-find, tile-local first (round 4) -------------------------------------------------------------------
// (measured, no gain -- see ws_zone_union.)  k_ws_union hooks every forward link of every voxel through
// global atomics (4.3 ms at 512^3, 28 ms at 1024^3).  Here a workgroup first closes a 64 x 8 x 8 block of voxels in LDS -- the
// same min-index hooking on a 4 096-entry local table -- and writes every member's local root (the block-local minimum, which
// is also the smallest linear index of its local set); a second launch then hooks only the links that LEAVE a block
// (1/64 of the x-links, 1/8 of the y- and z-links, and scipy's row / plane wrap-around links) through the global table.
// Final roots are the sets' minima either way: comp[] after k_ws_flatten is identical to the one-level version's.
constexpr int UX = 64, UY = 8, UZ = 8, UN = UX * UY * UZ;
__device__ __forceinline__ uint32_t ws_lfind(uint32_t *lp, uint32_t a) {
    uint32_t r = __hip_atomic_load(&lp[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (r != a) {
        a = r;
        r = __hip_atomic_load(&lp[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return a;
}
__device__ __forceinline__ void ws_lunite(uint32_t *lp, uint32_t a, uint32_t b) {
    for (;;) {
        a = ws_lfind(lp, a);
        b = ws_lfind(lp, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(&lp[b], a);
        if (old == b) return;
        b = old;
    }
}
// does the forward link k (14 .. 26) of the voxel at block-local (lx, ly, lz) / global (x, y, z) stay inside its block (and
// inside the volume as a LATTICE neighbour)?  Everything else -- links into another block, scipy's wrap-around neighbours --
// belongs to the second launch.  Both launches call this, so every link is taken exactly once.
__device__ __forceinline__ bool ws_link_is_local(const WsGeom &g, int k, int lx, int ly, int lz, int64_t x, int64_t y, int64_t z, int &lj) {
    const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
    const int lx2 = lx + dx, ly2 = ly + dy, lz2 = lz + dz;
    if (lx2 < 0 || lx2 >= UX || ly2 < 0 || ly2 >= UY || lz2 < 0 || lz2 >= UZ) return false;
    if (x + dx >= g.w || y + dy >= g.h || z + dz >= g.d) return false;
    lj = (lz2 * UY + ly2) * UX + lx2;
    return true;
}
__global__ __launch_bounds__(256) void k_ws_union_local(WsGeom g, const uint32_t *__restrict__ zmask, uint32_t *comp) {
    __shared__ uint32_t lp[UN];
    const int nux = (int)((g.w + UX - 1) / UX), nuy = (int)((g.h + UY - 1) / UY);
    const int64_t t = blockIdx.x;
    const int tx = (int)(t % nux);
    const int64_t r0 = t / nux;
    const int64_t x0 = (int64_t)tx * UX, y0 = (r0 % nuy) * UY, z0 = (r0 / nuy) * UZ;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t x = x0 + lane;
    constexpr int PER = UY * UZ / 4; // rows per wave
    uint32_t zm[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int r = wv * PER + i, ly = r % UY, lz = r / UY;
        const int64_t y = y0 + ly, z = z0 + lz;
        const uint32_t li = (uint32_t)(r * UX + lane);
        uint32_t m = 0;
        bool member = false;
        if (x < g.w && y < g.h && z < g.d) {
            const int64_t p = z * g.hw + y * g.w + x;
            member = comp[p] != ENTRY;
            if (member) m = (zmask[p] >> 14) | 0x80000000u; // (bit 31: "member", forward links in bits 0 .. 12)
        }
        lp[li] = member ? li : NONE;
        zm[i] = m;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; i++) {
        uint32_t m = zm[i] & 0x1FFFu;
        if (!m) continue;
        const int r = wv * PER + i, ly = r % UY, lz = r / UY;
        const uint32_t li = (uint32_t)(r * UX + lane);
        while (m) {
            const int k = 14 + __ffs(m) - 1;
            m &= m - 1;
            int lj;
            if (!ws_link_is_local(g, k, lane, ly, lz, x, y0 + ly, z0 + lz, lj)) continue;
            if (__hip_atomic_load(&lp[lj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == NONE) continue;
            ws_lunite(lp, li, (uint32_t)lj);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; i++) {
        if (!(zm[i] >> 31)) continue;
        const int r = wv * PER + i, ly = r % UY, lz = r / UY;
        const uint32_t root = ws_lfind(lp, (uint32_t)(r * UX + lane));
        const int rx = (int)(root % UX), rr = (int)(root / UX), ry = rr % UY, rz = rr / UY;
        comp[(z0 + lz) * g.hw + (y0 + ly) * g.w + x] = (uint32_t)((z0 + rz) * g.hw + (y0 + ry) * g.w + x0 + rx);
    }
}
__global__ __launch_bounds__(256) void k_ws_union_cross(WsGeom g, const uint32_t *__restrict__ zmask, uint32_t *comp) {
    const int64_t p64 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p64 >= g.n) return;
    const uint32_t p = (uint32_t)p64;
    if (__hip_atomic_load(&comp[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ENTRY) return;
    uint32_t zm = (zmask[p] >> 14) & 0x1FFFu;
    if (!zm) return;
    const uint32_t w = (uint32_t)g.w, hw = (uint32_t)g.hw; // (n < 2^32: 32-bit divisions)
    const uint32_t z = p / hw, rem = p - z * hw, y = rem / w, x = rem - y * w;
    const int lx = (int)(x % UX), ly = (int)(y % UY), lz = (int)(z % UZ);
    while (zm) {
        const int k = 14 + __ffs(zm) - 1;
        zm &= zm - 1;
        int lj;
        if (ws_link_is_local(g, k, lx, ly, lz, (int64_t)x, (int64_t)y, (int64_t)z, lj)) continue;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const int64_t q = (int64_t)p + dz * g.hw + dy * g.w + dx;
        if (__hip_atomic_load(&comp[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ENTRY) continue;
        ws_unite(comp, p, (uint32_t)q);
    }
}

__global__ __launch_bounds__(256) void k_ws_flatten(int64_t n, uint32_t *comp) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t c = comp[p];
    if (c == ENTRY || c == (uint32_t)p) return;
    comp[p] = ws_find(comp, c);
}

// zones / basins: comp[] from "ENTRY or self" to "ENTRY or the smallest index of the voxel's set"
static int ws_zone_union(const WsGeom &g, int conn, const uint32_t *zmask, uint32_t *comp, hipStream_t st) {
    // measured (profiles/r04_ws_union_ab.txt): block-local phase 1.85 ms + cross links 3.11 ms + flatten 0.73 ms against x-runs 0.55 +
    // one-level hooks 4.24 + flatten 0.86 ms at 512^3 -- the same 5.7 ms.  The 27 % of the links that leave a block are the ones
    // whose two ends lie on different cache lines (+y / +z neighbours), i.e. they were the cost all along; the links a block
    // closes in LDS shared lines anyway.  Same bits, same time: the one-level path stays the default, IVX_WS_UNION_LOCAL=1 opts in.
    static const bool two_level = []() { const char *e = getenv("IVX_WS_UNION_LOCAL"); return e && e[0] == '1'; }();
    const unsigned gl = (unsigned)cdiv(g.n, 256);
    if (two_level) {
        const int64_t nblk = cdiv(g.w, UX) * cdiv(g.h, UY) * cdiv(g.d, UZ);
        hipLaunchKernelGGL(k_ws_union_local, dim3((unsigned)nblk), dim3(256), 0, st, g, zmask, comp);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_ws_union_cross, dim3(gl), dim3(256), 0, st, g, zmask, comp);
        IVX_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_ws_runs, dim3(gl), dim3(256), 0, st, g, zmask, comp, (int)((g.smask >> 12) & 1u));
        IVX_LAUNCH_CHECK();
        WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_union<CC>, dim3(gl), dim3(256), 0, st, g, zmask, comp));
        IVX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_ws_flatten, dim3(gl), dim3(256), 0, st, g.n, comp);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// The cost map: rounds of dirty-tile visits until nothing changes (one host read per round through the mailbox).
template <bool SK>
static int ws_cost_rounds(const WsGeom &g, int conn, const uint16_t *I, uint16_t *C, uint32_t *list, uint8_t *dirty, uint8_t *pending,
                          WsState *wst, hipStream_t st, int64_t *rounds_out, int64_t *visits_out) {
    int64_t rounds = 0, visits = 0;
    IVX_HIP(hipMemsetAsync(pending, 0, (size_t)g.ntiles, st));
    hipLaunchKernelGGL(k_ws_wake, dim3(1), dim3(256), 0, st, (int64_t)0, dirty, pending, wst); // minrej = NONE
    IVX_LAUNCH_CHECK();
    const char *gate_env = getenv("IVX_WS_GATE");
    const bool gate = gate_env && gate_env[0] == '1'; // measured slower on the noise phantom (more rounds AND more visits): opt-in
    const bool trace = getenv("IVX_WS_TRACE") != nullptr;
    uint32_t theta = gate ? 0u : CINF;
    // Without the gate the host never stands between two rounds' kernels (round 6, as for the plateau rounds of the scikit-image
    // branch): a round's relaxation is launched on a grid guessed from the previous round's list (1.5 x + 64: the lists shrink
    // from round to round) BEFORE the host has read this round's length; a longer list gets a second launch behind it.
    static const bool ahead = []() { const char *e = getenv("IVX_WS_AHEAD"); return !(e && e[0] == '0'); }(); // (0: the list length is read before the launch -- A/B)
    if (!gate && ahead) {
        uint32_t guess = 0;
        int parity = 0; // (both list counters are zero here, and again when the loop ends on an empty list)
        for (;;) {
            uint32_t seq = 0, msg[3] = {0, 0, 0}, *cur = nullptr;
            int rc = ws_build_list_publish(g.ntiles, dirty, list, wst, parity, st, &seq, &cur);
            if (rc != IVX_OK) return rc;
            parity ^= 1;
            if (guess) {
                WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_ws_relax<CC, SK>), dim3(guess), dim3(256), 0, st, g, I, C, list, dirty, pending, wst, CINF | (trace ? 0x80000000u : 0u), 0u, cur));
                IVX_LAUNCH_CHECK();
            }
            rc = mailbox_wait(seq, st, msg, 3);
            if (rc != IVX_OK) return rc;
            const uint32_t nl = msg[0];
            if (trace) fprintf(stderr, "ws round %lld tiles %u (guessed grid %u) assigned %u\n", (long long)rounds, nl, guess, msg[2]);
            if (nl == 0) break; // nothing is refused without a gate: this is the fix-point
            rounds++;
            visits += nl;
            if (nl > guess) {
                WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_ws_relax<CC, SK>), dim3(nl - guess), dim3(256), 0, st, g, I, C, list, dirty, pending, wst, CINF | (trace ? 0x80000000u : 0u), guess, cur));
                IVX_LAUNCH_CHECK();
            }
            guess = (uint32_t)std::min<int64_t>(g.ntiles, (int64_t)nl + nl / 2 + 64);
            IVX_REQUIRE(rounds < 1000000, IVX_EHIP, "watershed: relaxation does not terminate");
        }
        *rounds_out = rounds;
        *visits_out = visits;
        return IVX_OK;
    }
    for (;;) {
        IVX_HIP(hipMemsetAsync(&wst->nlist, 0, 4, st));
        hipLaunchKernelGGL(k_ws_build_list, dim3((unsigned)cdiv(g.ntiles, 256 * BL_PER)), dim3(256), 0, st, g.ntiles, dirty, list, wst, &wst->nlist,
                           (uint32_t *)nullptr, (uint32_t *)nullptr, 0u);
        IVX_LAUNCH_CHECK();
        uint32_t seq = 0, msg[3] = {0, 0, 0};
        int rc = mailbox_publish(&wst->nlist, 3, st, &seq);
        if (rc != IVX_OK) return rc;
        rc = mailbox_wait(seq, st, msg, 3);
        if (rc != IVX_OK) return rc;
        const uint32_t nl = msg[0];
        if (trace) fprintf(stderr, "ws round %lld theta %u tiles %u minrej %u assigned %u\n", (long long)rounds, theta, nl, msg[1], msg[2]);
        if (nl == 0) {
            if (theta >= CINF || msg[1] == NONE) break; // nothing was refused: this is the fix-point
            // converged below the gate: lift it to the first level that has work, or all the way once the bulk is in
            theta = (uint64_t)msg[2] * 2 > (uint64_t)g.n ? CINF : msg[1];
            hipLaunchKernelGGL(k_ws_wake, dim3((unsigned)cdiv(g.ntiles, 256)), dim3(256), 0, st, g.ntiles, dirty, pending, wst);
            IVX_LAUNCH_CHECK();
            continue;
        }
        rounds++;
        visits += nl;
        WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_ws_relax<CC, SK>), dim3(nl), dim3(256), 0, st, g, I, C, list, dirty, pending, wst, theta | (trace ? 0x80000000u : 0u), 0u, &wst->nlist));
        IVX_LAUNCH_CHECK();
        IVX_REQUIRE(rounds < 1000000, IVX_EHIP, "watershed: relaxation does not terminate");
    }
    *rounds_out = rounds;
    *visits_out = visits;
    return IVX_OK;
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

} // namespace
