// k_ccl.hip -- region growing without a frontier: connected components of the candidate bit plane by run-based
// union-find, then "reached" = the components that hold a reached bit.
//
// Same result as the tile frontier of k_flood.hip (and therefore as generic_floodfill_threshold,
// invesalius_rs/src/floodfill.rs:96-237): for a SYMMETRIC structuring element whose centre row contains both x
// neighbours, the voxels the reference fills are exactly the candidate voxels connected to an in-range seed.
//
// Why: the frontier needs one dependent kernel per tile hop (about 20 hops across a 512^3 volume, ~15-20 us of tile
// latency each); union-find has no such chain -- every phase is one flat pass over the 16 MiB bit plane:
//   nodes      maximal runs of 1-bits inside each 64-bit word (a run that continues into the next word is two
//              nodes, joined by an ordinary union) -> no cross-word scanning anywhere
//   k_ccl_count / scan   runs per word -> node ids (wordbase + rank of the run's first bit)
//   k_ccl_union          one lane per word; each run is united with the runs it touches in the rows that PRECEDE it
//                        in raster order ((z-1, y-1..y+1), (z, y-1)) and with the run ending at bit 63 of the previous
//                        word; lock-free union by atomicMin on the larger root (stale parent reads only cost retries)
//   k_ccl_flatten        parent[i] = root(i)
//   k_ccl_activate       flag[root(run)] = 1 for every run that holds a reached bit (seeds, or halo bits that arrived
//                        from a Z-neighbour GPU)
//   k_ccl_paint          reached |= every run whose root is flagged
// All passes stream the bit planes (L2 / Infinity Cache resident) and the 4-byte-per-run parent table.
#include <stdlib.h>

#include <map>
#include <mutex>

#include "ivx_internal.h"

namespace {

struct CGeom {
    int64_t dz, dy, dx, wx, nwords;
    uint32_t strct;
};

__device__ __forceinline__ unsigned long long run_starts(unsigned long long c) { return c & ~(c << 1); }

// id of the run of word value c (whose runs start at id `base`) that contains bit p
__device__ __forceinline__ uint32_t run_id_at(unsigned long long c, uint32_t base, int p) {
    const unsigned long long st = run_starts(c);
    const unsigned long long upto = p == 63 ? ~0ull : ((2ull << p) - 1ull);
    const int s = 63 - __builtin_clzll(st & upto); // the run's first bit
    return base + (uint32_t)__popcll(st & ((1ull << s) - 1ull));
}
// last bit of the run containing bit p
__device__ __forceinline__ int run_end_at(unsigned long long c, int p) {
    const unsigned long long inv = ~(c >> p);
    return inv ? p + __builtin_ctzll(inv) - 1 : 63;
}

__global__ __launch_bounds__(256) void k_ccl_count(const unsigned long long *__restrict__ cand, int64_t nwords,
                                                   uint32_t *__restrict__ cnt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride)
        cnt[i] = (uint32_t)__popcll(run_starts(cand[i]));
}

// ---- exclusive scan of u32 (in place), three passes, 4096 elements per workgroup -------------------------------------
constexpr int SCAN_ITEMS = 16;
__global__ __launch_bounds__(256) void k_scan_block(uint32_t *__restrict__ data, int64_t n, uint32_t *__restrict__ bsum) {
    __shared__ uint32_t s_wave[4];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = base + k < n ? data[base + k] : 0u;
        sum += v[k];
    }
    uint32_t inc = sum;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t off = inc - sum;
    for (int q = 0; q < wv; q++) off += s_wave[q];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) data[base + k] = off;
        off += v[k];
    }
    if (threadIdx.x == 255) bsum[blockIdx.x] = off;
}
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t *__restrict__ bsum, int64_t nb, uint32_t *__restrict__ total) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
        const int64_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? bsum[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        uint32_t wb = 0;
        for (int q = 0; q < wv; q++) wb += s_wave[q];
        const uint32_t carry = s_carry;
        if (i < nb) bsum[i] = carry + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wb + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
__global__ __launch_bounds__(256) void k_scan_add(uint32_t *__restrict__ data, int64_t n, const uint32_t *__restrict__ bsum) {
    const uint32_t add = bsum[blockIdx.x];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) data[base + k] += add;
}

__global__ __launch_bounds__(256) void k_ccl_init(uint32_t *__restrict__ parent, uint8_t *__restrict__ flag, uint32_t nruns) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nruns; i += stride) {
        parent[i] = i;
        flag[i] = 0;
    }
}

// ---- union-find ------------------------------------------------------------------------------------------------------
// parent[x] <= x always and only ever decreases, and every value ever stored in parent[x] is a member of x's set.
// So plain (cacheable, possibly stale) loads are safe: a stale parent is a former ancestor, the walk still descends
// and terminates, and the only place that needs the truth -- "is a still a root?" -- is decided by the atomicMin.
// Path halving with plain stores is safe for the same reason (it only ever writes an ancestor).
// parent[x] <= x always, parent values only ever decrease, and every value ever stored in parent[x] is a member of
// x's set.  So plain (cacheable, possibly stale) loads are safe in find: a stale parent is a former ancestor, the
// walk still descends and terminates, and the only step that needs the truth -- "is a still a root?" -- is decided
// by the atomicMin in uf_union.  (Plain STORES into parent[] during the union are NOT safe -- measured: lost links.)
__device__ __forceinline__ uint32_t uf_find(const uint32_t *parent, uint32_t x) {
    for (;;) { // no compression here: halving through atomicMin was measured slower (0.96 vs 0.86 ms at 512^3)
        const uint32_t p = parent[x];
        if (p == x) return x;
        x = p;
    }
}
__device__ __forceinline__ uint32_t uf_find_ro(const uint32_t *parent, uint32_t x) {
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
        if (a < b) { const uint32_t t = a; a = b; b = t; } // link the larger root under the smaller
        const uint32_t old = atomicMin(&parent[a], b);
        if (old == a) return; // a was still a root: linked
        a = old;              // somebody re-parented a meanwhile: unite its new parent with b
    }
}


// ---- tile-local components in LDS ------------------------------------------------------------------------------------
// One workgroup per 64(x) x 16(y) x 16(z) tile (one word wide, lane = word), exactly the flood tile.  The runs of the
// tile are numbered locally (block scan), united through LDS atomics (cheap) over the adjacencies that stay inside the
// tile, flattened, and written out as parent[global id] = global id of the tile-local root.  Afterwards only the
// adjacencies that cross a tile face are left for the global union: a few global atomics per tile instead of one per
// run (global device-scope atomics were the entire cost of the untiled version: 1.6 ms at 512^3).
constexpr int LTW = 8, LTY = 8, LTZ = 4, LMAX = 256 * 32; // tile = 8 words (512 voxels) x 8 rows x 4 slices; <= 32 runs/word

__device__ __forceinline__ uint32_t l_find(uint32_t *lp, uint32_t x) {
    uint32_t p = lp[x];
    while (p != x) {
        const uint32_t gp = lp[p];
        if (gp != p) atomicMin(&lp[x], gp); // path halving; LDS atomics are cheap and keep parent[] monotone
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void l_union(uint32_t *lp, uint32_t a, uint32_t b) {
    for (;;) {
        a = l_find(lp, a);
        b = l_find(lp, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(&lp[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_ccl_local(const unsigned long long *__restrict__ cand,
                                                   const uint32_t *__restrict__ wbase, CGeom g, int64_t ntx, int64_t nty,
                                                   uint32_t *__restrict__ parent) {
    __shared__ uint32_t lp[LMAX];          // local parent
    __shared__ unsigned char lown[LMAX];   // owner lane of each local node
    __shared__ unsigned long long sC[256];
    __shared__ uint32_t sB[256], sG[256];
    __shared__ uint32_t s_wave[4];
    const int64_t tile = blockIdx.x;
    const int64_t txi = tile % ntx, r1 = tile / ntx;
    const int64_t tyi = r1 % nty, tzi = r1 / nty;
    const int lw = threadIdx.x & (LTW - 1), ly = (threadIdx.x >> 3) & (LTY - 1), lz = threadIdx.x >> 6;
    const int64_t w = txi * LTW + lw, y = tyi * LTY + ly, z = tzi * LTZ + lz;
    const bool inside = z < g.dz && y < g.dy && w < g.wx;
    const int64_t widx = inside ? (z * g.dy + y) * g.wx + w : 0;
    const unsigned long long c = inside ? cand[widx] : 0ull;
    const uint32_t nr = (uint32_t)__popcll(run_starts(c));
    // block exclusive scan of the run counts
    uint32_t inc = nr;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t lb = inc - nr;
    for (int q = 0; q < wv; q++) lb += s_wave[q];
    sC[threadIdx.x] = c;
    sB[threadIdx.x] = lb;
    sG[threadIdx.x] = inside ? wbase[widx] : 0u;
    for (uint32_t r = 0; r < nr; r++) {
        lp[lb + r] = lb + r;
        lown[lb + r] = (unsigned char)threadIdx.x;
    }
    __syncthreads();
    if (c) {
        unsigned long long rest = c;
        uint32_t lid = lb;
        while (rest) {
            const int a = __builtin_ctzll(rest);
            const int b = run_end_at(c, a);
            const unsigned long long rm = (b - a == 63) ? ~0ull : (((1ull << (b - a + 1)) - 1ull) << a);
            rest &= ~rm;
            // same row, previous word of the tile
            if (a == 0 && lw > 0 && (sC[threadIdx.x - 1] >> 63)) l_union(lp, lid, run_id_at(sC[threadIdx.x - 1], sB[threadIdx.x - 1], 63));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int oz = q < 3 ? -1 : 0, oy = q < 3 ? q - 1 : -1;
                const uint32_t m3 = (g.strct >> ((oz + 1) * 9 + (oy + 1) * 3)) & 7u;
                const int nz = lz + oz, ny = ly + oy;
                if (!m3 || nz < 0 || ny < 0 || ny >= LTY) continue; // that row is outside the tile: global pass
                const int nt = (nz * LTY + ny) * LTW + lw;
                const unsigned long long c1 = sC[nt];
                unsigned long long M = 0;
                if (m3 & 2u) M |= rm;
                if (m3 & 4u) M |= rm << 1;
                if (m3 & 1u) M |= rm >> 1;
                unsigned long long hit = c1 & M;
                while (hit) {
                    const int p = __builtin_ctzll(hit);
                    const int e = run_end_at(c1, p);
                    l_union(lp, lid, run_id_at(c1, sB[nt], p));
                    hit &= e == 63 ? 0ull : ~((2ull << e) - 1ull);
                }
                // diagonal contacts through the word boundary, when the neighbour word is in the tile
                if ((m3 & 4u) && b == 63 && lw + 1 < LTW && (sC[nt + 1] & 1ull)) l_union(lp, lid, run_id_at(sC[nt + 1], sB[nt + 1], 0));
                if ((m3 & 1u) && a == 0 && lw > 0 && (sC[nt - 1] >> 63)) l_union(lp, lid, run_id_at(sC[nt - 1], sB[nt - 1], 63));
            }
            lid++;
        }
    }
    __syncthreads();
    for (uint32_t r = 0; r < nr; r++) {
        const uint32_t root = l_find(lp, lb + r);
        const uint32_t ow = lown[root];
        parent[sG[threadIdx.x] + r] = sG[ow] + (root - sB[ow]);
    }
}

__global__ __launch_bounds__(256) void k_ccl_union(const unsigned long long *__restrict__ cand,
                                                   const uint32_t *__restrict__ wbase, CGeom g, uint32_t *parent,
                                                   int skip_local) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.nwords; i += stride) {
        const unsigned long long c = cand[i];
        if (!c) continue;
        const int64_t w = i % g.wx, row = i / g.wx;
        const int64_t y = row % g.dy, z = row / g.dy;
        const uint32_t base = wbase[i];
        // the four rows that precede (z, y) in raster order, with their x patterns (bit0: x+1 of ME touches, ...)
        unsigned long long nc[4][3]; // words w-1, w, w+1 of the neighbour row (0 when absent / not in strct)
        uint32_t nb[4][3];
        uint32_t m3s[4];
        bool same_tile[4]; // the (w, w) part of this adjacency was already united by k_ccl_local
        int64_t nidx[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int oz = q < 3 ? -1 : 0, oy = q < 3 ? q - 1 : -1;
            const uint32_t m3 = (g.strct >> ((oz + 1) * 9 + (oy + 1) * 3)) & 7u;
            const int64_t zz = z + oz, yy = y + oy;
            const bool ok = m3 && zz >= 0 && yy >= 0 && yy < g.dy;
            m3s[q] = ok ? m3 : 0u;
            same_tile[q] = skip_local && ok && (zz / LTZ == z / LTZ) && (yy / LTY == y / LTY); // row of the same tile
            nidx[q] = (zz * g.dy + yy) * g.wx + w;
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const bool has = ok && w + e - 1 >= 0 && w + e - 1 < g.wx;
                nc[q][e] = has ? cand[nidx[q] + e - 1] : 0ull;
                nb[q][e] = has ? wbase[nidx[q] + e - 1] : 0u;
            }
        }
        const unsigned long long pc = w > 0 ? cand[i - 1] : 0ull; // same row, previous word
        const uint32_t pb = w > 0 ? wbase[i - 1] : 0u;
        unsigned long long rest = c;
        uint32_t id = base;
        while (rest) {
            const int a = __builtin_ctzll(rest);
            const int b = run_end_at(c, a);
            const unsigned long long rm = (b - a == 63) ? ~0ull : (((1ull << (b - a + 1)) - 1ull) << a);
            rest &= ~rm;
            const bool wm_local = skip_local && (w % LTW) != 0;            // word w-1 belongs to the same tile
            const bool wp_local = skip_local && ((w + 1) % LTW) != 0;      // word w+1 belongs to the same tile
            if (a == 0 && (pc >> 63) && !wm_local) uf_union(parent, id, run_id_at(pc, pb, 63));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t m3 = m3s[q];
                if (!m3) continue;
                // my voxel x touches x+ox of the neighbour row for every ox in the pattern (ii = ox + 1)
                unsigned long long M = 0;
                if (m3 & 2u) M |= rm;
                if (m3 & 4u) M |= rm << 1;
                if (m3 & 1u) M |= rm >> 1;
                unsigned long long hit = same_tile[q] ? 0ull : (nc[q][1] & M);
                while (hit) {
                    const int p = __builtin_ctzll(hit);
                    const int e = run_end_at(nc[q][1], p);
                    uf_union(parent, id, run_id_at(nc[q][1], nb[q][1], p));
                    hit &= e == 63 ? 0ull : ~((2ull << e) - 1ull);
                }
                if ((m3 & 4u) && b == 63 && (nc[q][2] & 1ull) && !(same_tile[q] && wp_local))
                    uf_union(parent, id, run_id_at(nc[q][2], nb[q][2], 0));
                if ((m3 & 1u) && a == 0 && (nc[q][0] >> 63) && !(same_tile[q] && wm_local))
                    uf_union(parent, id, run_id_at(nc[q][0], nb[q][0], 63));
            }
            id++;
        }
    }
}

__global__ __launch_bounds__(256) void k_ccl_flatten(uint32_t *parent, uint32_t nruns) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nruns; i += stride) parent[i] = uf_find_ro(parent, i);
}

__global__ __launch_bounds__(256) void k_ccl_activate(const unsigned long long *__restrict__ cand,
                                                      const unsigned long long *__restrict__ reached,
                                                      const uint32_t *__restrict__ wbase, int64_t nwords,
                                                      const uint32_t *__restrict__ parent, uint8_t *__restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const unsigned long long c = cand[i];
        unsigned long long hit = reached[i] & c;
        while (hit) {
            const int p = __builtin_ctzll(hit);
            const int e = run_end_at(c, p);
            flag[parent[run_id_at(c, wbase[i], p)]] = 1;
            hit &= e == 63 ? 0ull : ~((2ull << e) - 1ull);
        }
    }
}

__global__ __launch_bounds__(256) void k_ccl_paint(const unsigned long long *__restrict__ cand,
                                                   unsigned long long *__restrict__ reached,
                                                   const uint32_t *__restrict__ wbase, int64_t nwords,
                                                   const uint32_t *__restrict__ parent, const uint8_t *__restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const unsigned long long c = cand[i];
        if (!c) continue;
        unsigned long long rest = c, out = 0;
        uint32_t id = wbase[i];
        while (rest) {
            const int a = __builtin_ctzll(rest);
            const int b = run_end_at(c, a);
            const unsigned long long rm = (b - a == 63) ? ~0ull : (((1ull << (b - a + 1)) - 1ull) << a);
            rest &= ~rm;
            if (flag[parent[id]]) out |= rm;
            id++;
        }
        if (out) reached[i] |= out;
    }
}

// voxels per component: every run adds its length to its root.  The wave first agrees on the root its first lane's run
// has (background components own most runs of a wave) and adds that part with ONE atomic.
__global__ __launch_bounds__(256) void k_ccl_sizes(const unsigned long long *__restrict__ cand,
                                                   const uint32_t *__restrict__ wbase, int64_t nwords,
                                                   const uint32_t *__restrict__ parent, uint32_t *__restrict__ sizes) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < nwords; base += stride) {
        const int64_t i = base + threadIdx.x;
        const unsigned long long c = i < nwords ? cand[i] : 0ull;
        unsigned long long rest = c;
        uint32_t id = i < nwords ? wbase[i] : 0u;
        const unsigned long long havem = __ballot(c != 0ull);
        if (!havem) continue;
        const uint32_t hot = __shfl(c ? parent[id] : 0u, __builtin_ctzll(havem), 64); // root of the wave's first run
        uint32_t hot_add = 0;
        while (rest) {
            const int a = __builtin_ctzll(rest);
            const int b = run_end_at(c, a);
            const unsigned long long rm = (b - a == 63) ? ~0ull : (((1ull << (b - a + 1)) - 1ull) << a);
            rest &= ~rm;
            const uint32_t root = parent[id];
            if (root == hot) hot_add += (uint32_t)(b - a + 1);
            else atomicAdd(&sizes[root], (uint32_t)(b - a + 1));
            id++;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hot_add += __shfl_xor(hot_add, o, 64);
        if ((threadIdx.x & 63) == 0 && hot_add) atomicAdd(&sizes[hot], hot_add);
    }
}
__global__ __launch_bounds__(256) void k_ccl_flag_small(const uint32_t *__restrict__ sizes, uint32_t nruns, uint32_t max_size,
                                                        uint8_t *__restrict__ flag, int *__restrict__ any_small) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nruns; i += stride) {
        const uint32_t s = sizes[i]; // non-zero only at roots
        const bool small = s > 0 && s <= max_size;
        flag[i] = small ? 1 : 0;
        if (small) *any_small = 1; // benign race: all write 1
    }
}

// host-side state per scratch buffer: is the component table valid for the current candidate plane?
struct CclState {
    bool built = false;
    uint32_t nruns = 0;
};
static std::map<const void *, CclState> g_state;
static std::map<hipStream_t, const void *> g_tab_owner; // the tables live in per-stream workspaces: ONE scratch per stream at a time
static std::mutex g_state_mu;

static inline int grid_for(int64_t n) {
    const int64_t b = ivx::cdiv(n, 256);
    return (int)(b < 1 ? 1 : (b < 16384 ? b : 16384));
}

} // namespace

namespace ivx {

void ccl_invalidate(const void *scratch) {
    std::lock_guard<std::mutex> lk(g_state_mu);
    g_state[scratch] = CclState();
}

void ccl_forget_stream(void *stream) {
    std::lock_guard<std::mutex> lk(g_state_mu);
    auto it = g_tab_owner.find((hipStream_t)stream);
    if (it == g_tab_owner.end()) return;
    g_state.erase(it->second); // its tables lived in the stream's workspaces, which are being freed
    g_tab_owner.erase(it);
}

bool ccl_supported(uint32_t strct_bits) {
    const uint32_t s = strct_bits | (1u << 13);
    for (int k = 0; k < 27; k++)
        if (((s >> k) & 1u) != ((s >> (26 - k)) & 1u)) return false; // symmetric under point reflection
    return (s >> 12 & 1u) && (s >> 14 & 1u);                      // x neighbours in the centre row: runs are connected
}

struct CclTables {
    CGeom g;
    uint32_t *wbase = nullptr, *parent = nullptr;
    uint8_t *flag = nullptr;
    uint32_t nruns = 0;
};

// Component table of the candidate plane (run ids per word, flattened union-find), built on first use per scratch key.
// extra_per_run: additional bytes per run reserved behind parent[] and flag[] (the caller's own per-run arrays).
static int ccl_prepare(const ivx_flood_plan *p, const uint64_t *cand, const void *scratch_key, hipStream_t st, CclTables *t,
                       size_t extra_per_run = 0) {
    CGeom &g = t->g;
    g.dz = p->dz; g.dy = p->dy; g.dx = p->dx; g.wx = p->wx;
    g.nwords = p->dz * p->dy * p->wx;
    g.strct = p->strct_bits | (1u << 13);
    if (g.nwords == 0) return IVX_OK;
    IVX_REQUIRE(g.nwords < 0x7fffffffll * 2, IVX_EINVAL, "flood: volume too large for 32-bit run ids");
    CclState state;
    {
        std::lock_guard<std::mutex> lk(g_state_mu);
        state = g_state[scratch_key];
        if (g_tab_owner[st] != scratch_key) state = CclState(); // another flood on this stream rebuilt the tables meanwhile
        g_tab_owner[st] = scratch_key;
    }
    const int64_t nsb = cdiv(g.nwords, 256 * SCAN_ITEMS);
    void *d_wbase, *d_tab;
    int rc;
    if ((rc = ws_get_s(WS_CCL0, st, (size_t)g.nwords * 4 + (size_t)nsb * 4 + 64, &d_wbase))) return rc;
    uint32_t *wbase = (uint32_t *)d_wbase;
    uint32_t *bsum = wbase + g.nwords;
    uint32_t *d_total = bsum + nsb;
    const unsigned long long *c = (const unsigned long long *)cand;
    if (!state.built) {
        hipLaunchKernelGGL(k_ccl_count, dim3(grid_for(g.nwords)), dim3(256), 0, st, c, g.nwords, wbase);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nsb), dim3(256), 0, st, wbase, g.nwords, bsum);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, bsum, nsb, d_total);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nsb), dim3(256), 0, st, wbase, g.nwords, bsum);
        IVX_LAUNCH_CHECK();
        uint32_t seq, tot = 0;
        if ((rc = mailbox_publish(d_total, 1, st, &seq))) return rc;
        if ((rc = mailbox_wait(seq, st, &tot, 1))) return rc;
        state.nruns = tot;
        IVX_REQUIRE(tot < 0xfffffff0u, IVX_EINVAL, "flood: too many runs for 32-bit ids");
    }
    t->wbase = wbase;
    t->nruns = state.nruns;
    if (state.nruns == 0) {
        std::lock_guard<std::mutex> lk(g_state_mu);
        state.built = true;
        g_state[scratch_key] = state;
        return IVX_OK;
    }
    if ((rc = ws_get_s(WS_CCL1, st, (size_t)state.nruns * (5 + extra_per_run) + 256, &d_tab))) return rc;
    uint32_t *parent = (uint32_t *)d_tab;
    uint8_t *flag = (uint8_t *)(parent + state.nruns);
    t->parent = parent;
    t->flag = flag;
    if (!state.built) {
        hipLaunchKernelGGL(k_ccl_init, dim3(grid_for(state.nruns)), dim3(256), 0, st, parent, flag, state.nruns);
        IVX_LAUNCH_CHECK();
        const int64_t ntx = cdiv(g.wx, LTW), nty = cdiv(g.dy, LTY), ntz = cdiv(g.dz, LTZ);
        IVX_REQUIRE(ntx * nty * ntz < 0x7fffffffll, IVX_EINVAL, "flood: too many tiles");
        hipLaunchKernelGGL(k_ccl_local, dim3((unsigned)(ntx * nty * ntz)), dim3(256), 0, st, c, wbase, g, ntx, nty, parent);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_ccl_union, dim3(grid_for(g.nwords)), dim3(256), 0, st, c, wbase, g, parent, 1);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_ccl_flatten, dim3(grid_for(state.nruns)), dim3(256), 0, st, parent, state.nruns);
        IVX_LAUNCH_CHECK();
        std::lock_guard<std::mutex> lk(g_state_mu);
        state.built = true;
        g_state[scratch_key] = state;
    }
    return IVX_OK;
}

// reached := union of the candidate components that hold a reached bit.  Builds the component table on first use.
int ccl_run(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, const void *scratch_key, hipStream_t st) {
    CclTables t;
    int rc = ccl_prepare(p, cand, scratch_key, st, &t);
    if (rc) return rc;
    if (t.g.nwords == 0 || t.nruns == 0) return IVX_OK;
    const unsigned long long *c = (const unsigned long long *)cand;
    hipLaunchKernelGGL(k_ccl_activate, dim3(grid_for(t.g.nwords)), dim3(256), 0, st, c, (const unsigned long long *)reached,
                       t.wbase, t.g.nwords, t.parent, t.flag);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ccl_paint, dim3(grid_for(t.g.nwords)), dim3(256), 0, st, c, (unsigned long long *)reached, t.wbase,
                       t.g.nwords, t.parent, t.flag);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// out := the voxels of every candidate component of at most max_size voxels (out must be zeroed by the caller);
// *any_small (device int) is raised when there is one.  The component sizes are the sums of the run lengths per root.
int ccl_small_components(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *out, uint32_t max_size, int *any_small,
                         const void *scratch_key, hipStream_t st) {
    CclTables t;
    ccl_invalidate(scratch_key); // the flags are reused with another meaning: never trust a table built for a flood
    int rc = ccl_prepare(p, cand, scratch_key, st, &t, 4);
    if (rc) return rc;
    if (t.g.nwords == 0 || t.nruns == 0) return IVX_OK;
    uint32_t *sizes = (uint32_t *)(((uintptr_t)(t.flag + t.nruns) + 15) & ~(uintptr_t)15);
    const unsigned long long *c = (const unsigned long long *)cand;
    IVX_HIP(hipMemsetAsync(sizes, 0, (size_t)t.nruns * 4, st));
    hipLaunchKernelGGL(k_ccl_sizes, dim3(grid_for(t.g.nwords)), dim3(256), 0, st, c, t.wbase, t.g.nwords, t.parent, sizes);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ccl_flag_small, dim3(grid_for(t.nruns)), dim3(256), 0, st, sizes, t.nruns, max_size, t.flag, any_small);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ccl_paint, dim3(grid_for(t.g.nwords)), dim3(256), 0, st, c, (unsigned long long *)out, t.wbase,
                       t.g.nwords, t.parent, t.flag);
    IVX_LAUNCH_CHECK();
    ccl_invalidate(scratch_key);
    return IVX_OK;
}

} // namespace ivx
