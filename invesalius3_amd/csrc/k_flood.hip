// k_flood.hip -- seeded region growing (connected threshold flood) on bit planes.
//
// Reference semantics (bit-exact):
//   generic_floodfill_threshold          invesalius_rs/src/floodfill.rs:96-166
//   generic_floodfill_threshold_inplace  invesalius_rs/src/floodfill.rs:168-237
//   floodfill_internal                   invesalius_rs/src/floodfill.rs:5-49       (data == v, forced seed: ivx_floodfill)
//   floodfill_auto_threshold             invesalius_rs/src/floodfill_py.rs:12-85   (directed: edge planes, tile_update_dir)
// The reference walks a LIFO stack; the set it fills does not depend on the visiting order:
//   filled = connected component (under `strct`) of the in-range seeds inside
//            C = { v : t0 <= data[v] <= t1  and  barrier[v] != fill }   (+ the in-range seeds themselves),
// so a parallel fix-point gives the same bytes.
//
// MI355X design (memory-bound, no MFMA):
//   k_flood_candidates16  ONE streaming pass over data (+ barrier): 2 B + 1 B read per voxel, 1 bit written (lane = 16
//                       voxels).  C is 1 bit/voxel (64 voxels of an x-row per uint64): 16 MiB at 512^3, i.e. it lives
//                       in L2 / Infinity Cache for the whole flood.
//   k_flood_round_list  one workgroup per DIRTY tile of 64(x) x 16(y) x 16(z) voxels = 256 words, one lane per word,
//                       tiles taken from a compact per-round list.  The reached bits of the tile plus a one-row halo
//                       are staged in LDS together with their x-dilated twins (tile_update) and iterated to the
//                       tile-local fix-point:
//                         - 26/18/6-neighbour gather = OR of <= 9 LDS words, a compile-time pattern for the three
//                           standard structures so the reads issue back to back,
//                         - propagation ALONG x inside a word is closed in O(1) with the carry trick
//                           ((C + R) ^ C) & C  (and its bit-reversed twin), i.e. a 64-voxel run fills in one step,
//                         - one s_barrier per iteration (wave ballot + LDS flag vote).
//                       A tile whose boundary changed enlists exactly the neighbour tiles that can see the change
//                       (byte flags de-duplicate); every round reports its list length to a pinned progress line as it
//                       starts and the host keeps a few rounds queued ahead of the newest one it has seen.  Global
//                       rounds are bounded by the number of TILES on the longest path, not voxels; past 48 rounds the
//                       union-find engine (k_ccl.hip) takes over.
//   k_flood_coarse      before the rounds: tiles that are ALL candidate are flooded as units on the tile graph (one
//                       workgroup, rows of tile bits), so the solid interior of a region costs no rounds at all.
//   k_flood_persistent  the same tile update in ONE launch with a device-side ticket queue (opt-in, measured slower).
//   k_flood_apply(2)    reached bits -> out[v] = fill (only words with reached bits touch memory).
#include <math.h>
#include <map>
#include <mutex>
#include <stdlib.h>

#include "ivx_internal.h"

typedef short short8_t __attribute__((ext_vector_type(8)));

namespace {

constexpr int TY_LOG = 4, TY = 1 << TY_LOG, TZ = 16; // tile rows / slices (tile is one 64-voxel word wide)
constexpr int NT = TY * TZ;           // lanes per tile workgroup (one per word)
constexpr int HY = TY + 2, HZ = TZ + 2;
constexpr int BATCH = 8;              // most rounds the host may keep queued ahead (the counter ring has 2 * BATCH entries)
constexpr int SUB = 1;                // gather/update steps per termination vote (measured: 4 doubles the tile time)
// dirty-byte bits: bit 0 = "on (or to be put on) a round's list"; CLOSED = every candidate of the tile is reached, so no
// news can ever change it again (sticky for the rest of the flood; the byte is then never 0 and the enlisting atomicOr of a
// neighbour finds it "already marked": closed tiles cost no visits -- about a third of the second round's list on the bench
// volume); ENLISTED (coarse pass) further down
constexpr unsigned int CLOSED = 0x40u;
// (Tried in round 3: the round's list as 8 sublists with a counter each on its own 128-byte line.  ~10^3 appends that arrive
// together serialise on one counter at 5 - 10 ns each -- tools/micro/atomic_tail.hip: 1 024 workgroups x 1 append cost 5.5 us
// more than none, x 4 appends 40 us more, spread over 8 counters 0.3 / 1.3 us -- but the rounds' appends do not arrive
// together: region growing 0.192 -> 0.195 ms, the IFT cost levels 13.8 -> 14.3 ms.  Dropped.)

struct Tiles {
    int64_t dz, dy, dx, wx;
    int64_t nty, ntz, ntiles;
    uint32_t strct;
    int itcap; // local iterations per tile visit before the tile re-enlists itself
    int conn; // 6 / 18 / 26 when strct is exactly scipy's generate_binary_structure(3, 1|2|3), else 0 (generic path)
};

static int make_tiles(const ivx_flood_plan *p, Tiles *t) {
    IVX_REQUIRE(p && p->dz >= 0 && p->dy >= 0 && p->dx >= 0, IVX_EINVAL, "flood: bad shape");
    IVX_REQUIRE(p->wx == ivx::cdiv(p->dx, 64), IVX_EINVAL, "flood: plan.wx must be ceil(dx/64)");
    t->dz = p->dz; t->dy = p->dy; t->dx = p->dx; t->wx = p->wx;
    t->nty = ivx::cdiv(p->dy, TY); t->ntz = ivx::cdiv(p->dz, TZ);
    t->ntiles = t->wx * t->nty * t->ntz;
    t->strct = p->strct_bits & ~(1u << 13); // the centre never matters
    {
        uint32_t m6 = 0, m18 = 0, m26 = 0;
        for (int k = 0; k < 27; k++) {
            const int nzc = (k / 9 != 1) + ((k / 3) % 3 != 1) + (k % 3 != 1);
            if (nzc <= 1) m6 |= 1u << k;
            if (nzc <= 2) m18 |= 1u << k;
            m26 |= 1u << k;
        }
        const uint32_t sb = p->strct_bits | (1u << 13);
        t->conn = sb == m26 ? 26 : sb == m18 ? 18 : sb == m6 ? 6 : 0;
    }
    if (getenv("IVX_FLOOD_DBG")) t->strct |= 1u << 30;
    // One tile crossing (TY = TZ = 16 rows) per visit: a tile that is still changing after that re-enlists itself and
    // carries on in the next round, when its neighbours have already started on what it has published so far -- the
    // rounds pipeline instead of waiting for the slowest tile's local fix-point (measured 0.240 -> 0.232 ms on the bench
    // volume; below 16 a straight crossing needs two visits and the round count doubles).
    static const int itcap = [] {
        const char *e = getenv("IVX_FLOOD_ITCAP");
        const int v = e ? atoi(e) : TY + TY / 2; // (round 6: 24 instead of 16 -- 9 rounds instead of 10 on the bench volume, 0.1916 -> 0.1871 ms; 20 .. 64 measure alike)
        return v < 1 ? 1 : v;
    }();
    t->itcap = itcap;
    IVX_REQUIRE(t->ntiles < 0x7fffffffll, IVX_EINVAL, "flood: too many tiles");
    return IVX_OK;
}

// scratch: dirty[2][ntiles] u8 | counter ring | persistent-frontier queue | seed staging | round lists | coarse-pass row words
constexpr size_t SEED_CHUNK = 4096;
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct FScratch {
    size_t off_dirty0, off_dirty1, off_cnt, off_queue, off_queued, off_seeds, off_status, off_ring, off_list0, off_list1;
    size_t off_full, off_whole, total;
    uint32_t qcap;
};
// persistent-frontier queue header (device)
struct Queue { // one 128-B line per hot word: tickets, pushes, in-flight count and the read-mostly done flag
    unsigned int head, pad0[31];
    unsigned int tail, pad1[31];
    unsigned int pending, pad2[31];
    unsigned int done, abort, visits, pad3[29];
};
constexpr unsigned int Q_EMPTY = 0xffffffffu;
static FScratch make_fscratch(const Tiles &t) {
    FScratch s;
    s.off_dirty0 = 0;
    s.off_dirty1 = al256((size_t)t.ntiles);
    s.off_cnt = al256(s.off_dirty1 + (size_t)t.ntiles);
    s.off_queue = al256(s.off_cnt + 64 * 4);
    s.off_queued = al256(s.off_queue + sizeof(Queue));
    s.off_seeds = al256(s.off_queued + (size_t)t.ntiles * 4); // everything before off_seeds is zeroed by flood_clear
    s.off_status = al256(s.off_seeds + SEED_CHUNK * 3 * 8);
    s.off_ring = al256(s.off_status + 64);
    uint32_t q = 64;
    while ((int64_t)q < 2 * t.ntiles + 2048) q <<= 1; // <= ntiles queued + <= 1024 waiting tickets, 2x slack
    s.qcap = q;
    s.off_list0 = al256(s.off_ring + (size_t)q * 4);
    s.off_list1 = al256(s.off_list0 + (size_t)t.ntiles * 4);
    s.off_full = al256(s.off_list1 + (size_t)t.ntiles * 4); // coarse pass: one word per row of tiles, all-candidate / wholly reached
    s.off_whole = al256(s.off_full + (size_t)(t.nty * t.ntz) * 8);
    s.total = al256(s.off_whole + (size_t)(t.nty * t.ntz) * 8);
    return s;
}

template <typename T> __device__ __forceinline__ double as_double(T v) { return (double)v; }

// ---- candidates: one lane per output byte (8 voxels) ------------------------------------------------
// BAR: 0 none, 1 uint8 barrier array (out != fill), 2 in-place (data value != fill)
// fast path: int16/uint16 data, dx % 16 == 0, 16-B aligned pointers: one lane = 16 voxels (2 x 16-B data loads,
// one 16-B barrier load, one 2-B store), the same access shape as the threshold kernel
template <typename T, int BAR>
__global__ __launch_bounds__(256) void k_flood_candidates16(const short8_t *__restrict__ data,
                                                            const uint4 *__restrict__ bar, int64_t nchunks, double t0,
                                                            double t1, double fill, uint16_t *__restrict__ cand) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int lo = (int)ceil(t0 < -70000.0 ? -70000.0 : t0), hi = (int)floor(t1 > 70000.0 ? 70000.0 : t1); // integer data
    const unsigned fb = (unsigned)(uint8_t)fill;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += stride) {
        const short8_t a = __builtin_nontemporal_load(&data[2 * c]);
        const short8_t b = __builtin_nontemporal_load(&data[2 * c + 1]);
        uint4 q = make_uint4(0, 0, 0, 0);
        if (BAR == 1) q = bar[c];
        const unsigned bw[4] = {q.x, q.y, q.z, q.w};
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int v = (int)(T)(e < 8 ? a[e & 7] : b[e & 7]);
            bool ok = v >= lo && v <= hi;
            if (BAR == 1) ok = ok && ((bw[e >> 2] >> (8 * (e & 3))) & 0xffu) != fb;
            if (BAR == 2) ok = ok && (double)v != fill;
            m |= ok ? (1u << e) : 0u;
        }
        cand[c] = (uint16_t)m;
    }
}

template <typename T, int BAR>
__global__ __launch_bounds__(256) void k_flood_candidates(const T *__restrict__ data, const uint8_t *__restrict__ bar,
                                                          Tiles t, double t0, double t1, double fill,
                                                          uint8_t *__restrict__ cand) {
    const int64_t bpr = t.wx * 8; // bytes per bit-row
    const int64_t total = t.dz * t.dy * bpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool vec = (sizeof(T) == 2) && (t.dx % 8 == 0) && (((uintptr_t)data & 15) == 0) &&
                     (BAR != 1 || ((uintptr_t)bar & 7) == 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / bpr, q = i - row * bpr;
        const int64_t x0 = q * 8;
        unsigned m = 0;
        if (x0 < t.dx) {
            const int64_t base = row * t.dx + x0;
            if (vec) { // x0+8 <= dx because dx % 8 == 0
                const short8_t v = *reinterpret_cast<const short8_t *>(data + base);
                unsigned long long b8 = 0;
                if (BAR == 1) b8 = *reinterpret_cast<const unsigned long long *>(bar + base);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const double d = (double)(T)v[e];
                    bool ok = d >= t0 && d <= t1;
                    if (BAR == 1) ok = ok && (((b8 >> (8 * e)) & 0xff) != (unsigned long long)(uint8_t)fill);
                    if (BAR == 2) ok = ok && d != fill;
                    m |= ok ? (1u << e) : 0u;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    if (x0 + e < t.dx) {
                        const double d = as_double(data[base + e]);
                        bool ok = d >= t0 && d <= t1;
                        if (BAR == 1) ok = ok && bar[base + e] != (uint8_t)fill;
                        if (BAR == 2) ok = ok && d != fill;
                        m |= ok ? (1u << e) : 0u;
                    }
                }
            }
        }
        cand[i] = (uint8_t)m;
    }
}

// ---- seeds ----------------------------------------------------------------------------------------
// Bits that enter `reached` from OUTSIDE a tile visit (seeds, a neighbour slab's plane OR-ed into a halo slice) are news
// nobody has reported: a visit only tells its neighbours about the faces IT changed.  Wake the word's own tile and every
// tile that can see the word.
__device__ __forceinline__ void mark_tile_nbhd(const Tiles &t, uint8_t *__restrict__ dirty, int64_t tz, int64_t ty, int64_t tx) {
    for (int64_t az = tz - 1; az <= tz + 1; az++)
        for (int64_t ay = ty - 1; ay <= ty + 1; ay++)
            for (int64_t ax = tx - 1; ax <= tx + 1; ax++)
                if (az >= 0 && az < t.ntz && ay >= 0 && ay < t.nty && ax >= 0 && ax < t.wx)
                    dirty[(az * t.nty + ay) * t.wx + ax] = 1;
}

template <typename T>
__global__ void k_flood_seed(const T *__restrict__ data, Tiles t, double t0, double t1, const int64_t *__restrict__ seeds,
                             int64_t nseeds, unsigned long long *__restrict__ cand,
                             unsigned long long *__restrict__ reached, uint8_t *__restrict__ dirty) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nseeds) return;
    const int64_t x = seeds[3 * n], y = seeds[3 * n + 1], z = seeds[3 * n + 2];
    const double v = as_double(data[(z * t.dy + y) * t.dx + x]);
    if (!(v >= t0 && v <= t1)) return; // floodfill.rs:123: only in-range seeds start a flood
    const int64_t w = (z * t.dy + y) * t.wx + (x >> 6);
    const unsigned long long bit = 1ull << (x & 63);
    atomicOr(&cand[w], bit); // a seed expands even when its barrier byte already equals `fill`
    atomicOr(&reached[w], bit);
    mark_tile_nbhd(t, dirty, z / TZ, y / TY, x >> 6);
}

struct SeedPack {
    int64_t xyz[16][3];
    int n;
};
template <typename T>
__global__ void k_flood_seed_args(const T *__restrict__ data, Tiles t, double t0, double t1, SeedPack sp,
                                  unsigned long long *__restrict__ cand, unsigned long long *__restrict__ reached,
                                  uint8_t *__restrict__ dirty) {
    const int n = threadIdx.x;
    if (n >= sp.n) return;
    const int64_t x = sp.xyz[n][0], y = sp.xyz[n][1], z = sp.xyz[n][2];
    const double v = as_double(data[(z * t.dy + y) * t.dx + x]);
    if (!(v >= t0 && v <= t1)) return;
    const int64_t w = (z * t.dy + y) * t.wx + (x >> 6);
    const unsigned long long bit = 1ull << (x & 63);
    atomicOr(&cand[w], bit);
    atomicOr(&reached[w], bit);
    mark_tile_nbhd(t, dirty, z / TZ, y / TY, x >> 6);
}

// ---- one round over the dirty tiles ---------------------------------------------------------------
__device__ __forceinline__ unsigned long long fill_up(unsigned long long seed, unsigned long long m) {
    return (((m + seed) ^ m) & m) | seed;
}
__device__ __forceinline__ unsigned long long fill_runs(unsigned long long seed, unsigned long long m) {
    unsigned long long f = fill_up(seed, m);
    return __brevll(fill_up(__brevll(f), __brevll(m)));
}

// Tile update shared by the per-round kernel and the persistent frontier.
// LDS holds, for the 18x18 rows of the tile + halo: sN = the row's word, sD = the row's word dilated by one voxel
// along x including the carry bits of the x-neighbour words (so a full 3-wide strct row costs ONE LDS read), and the
// two carry bits themselves (generic structuring elements).  The local fix-point is a chaotic relaxation: reached
// bits only ever get OR'ed in, so a lane may read a neighbour's word while it is being rewritten -- any mix of old
// and new bits is a valid intermediate state -- and ONE barrier per iteration (the termination vote) is enough.
// ATOMIC: stage / publish with agent-scope atomics (needed when other workgroups update neighbours concurrently
// AND no kernel boundary follows, i.e. in the persistent frontier).
__device__ unsigned long long g_dbg[16];
#define DBG_T(i) if (DBG && threadIdx.x == 0) g_dbg[i] = __builtin_readcyclecounter();
// workgroup barrier that only waits for this wave's LDS traffic (lgkmcnt), NOT for its global stores/atomics in flight:
// __syncthreads() also drains vmcnt, which would park the whole tile behind the publish stores (~2.5 us).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0), vmcnt/expcnt untouched
    __builtin_amdgcn_s_barrier();
}
struct TileLds {
    unsigned long long sN[HZ * HY];
    unsigned long long sD[HZ * HY];
    unsigned char sCL[HZ * HY], sCR[HZ * HY];
    unsigned int dirs;
    unsigned int open; // some candidate of the tile is still unreached after this visit
    unsigned int vote[2]; // workgroup "anything changed" vote, double-buffered: ONE s_barrier per iteration
};

template <bool ATOMIC, int CONN, bool DBG = false>
__device__ __forceinline__ void tile_update(const Tiles &t, const unsigned long long *__restrict__ cand,
                                            unsigned long long *reached, int64_t tile, TileLds &L) {
    const int64_t txi = tile % t.wx, r1 = tile / t.wx;
    const int64_t tyi = r1 % t.nty, tzi = r1 / t.nty;
    const int64_t z0 = tzi * TZ, y0 = tyi * TY;
    // stage: lane idx <-> halo row idx (324 rows = 2 rows per lane for the first 68 lanes); 3 words per row; all six
    // loads of a lane are issued before any of them is consumed
    unsigned long long w3[2][3] = {{0ull, 0ull, 0ull}, {0ull, 0ull, 0ull}};
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int idx = threadIdx.x + pass * NT;
        if (idx < HZ * HY) {
            const int yy = idx % HY, zz = idx / HY;
            const int64_t z = z0 + zz - 1, y = y0 + yy - 1;
            if (z >= 0 && z < t.dz && y >= 0 && y < t.dy) {
                unsigned long long *row = reached + (z * t.dy + y) * t.wx;
#pragma unroll
                for (int xx = 0; xx < 3; xx++) {
                    const int64_t w = txi + xx - 1;
                    if (w >= 0 && w < t.wx)
                        w3[pass][xx] = ATOMIC ? __hip_atomic_load(row + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : row[w];
                }
            }
        }
    }
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int idx = threadIdx.x + pass * NT;
        if (idx < HZ * HY) {
            const unsigned long long cl = w3[pass][0] >> 63, cr = w3[pass][2] & 1ull, n = w3[pass][1];
            L.sN[idx] = n;
            L.sD[idx] = n | (n << 1) | (n >> 1) | cl | (cr << 63);
            L.sCL[idx] = (unsigned char)cl;
            L.sCR[idx] = (unsigned char)cr;
        }
    }
    const int ty = threadIdx.x & (TY - 1), tz = threadIdx.x >> TY_LOG;
    const int64_t z = z0 + tz, y = y0 + ty;
    const bool inside = z < t.dz && y < t.dy;
    const unsigned long long c = inside ? cand[(z * t.dy + y) * t.wx + txi] : 0ull;
    if (threadIdx.x == 0) L.vote[0] = 0u;
    __syncthreads();
    DBG_T(1)
    const int me = (tz + 1) * HY + (ty + 1);
    const unsigned long long r_in = L.sN[me];
    const unsigned long long mycl = L.sCL[me], mycr = L.sCR[me];
    unsigned long long r = r_in;
    const uint32_t st = t.strct;
    const bool xrun = (st >> 12 & 1) && (st >> 14 & 1); // both x neighbours of the centre row present
    bool exhausted = true;
    for (int it = 0; it < t.itcap; it++) {
      bool changed = false;
      // SUB gather/update steps per vote: LDS is coherent inside the workgroup and bits are only OR'ed in, so steps need
      // no barrier between them -- a wave always sees its own rows' previous step (program order), other waves' rows
      // whenever they land.  Only the termination vote needs the barrier.
#pragma unroll
      for (int sub = 0; sub < SUB; sub++) {
        unsigned long long nb = 0;
        if (CONN != 0) {
            // standard 6 / 18 / 26 structures: the row pattern is known at compile time, so the nine LDS reads are
            // issued back to back (no scalar branches between them) and OR'ed once they land
            unsigned long long v[9];
#pragma unroll
            for (int kk = 0; kk < 3; kk++)
#pragma unroll
                for (int jj = 0; jj < 3; jj++) {
                    const int nz = (kk != 1) + (jj != 1); // non-zero offsets among (dz, dy)
                    const int src = (tz + 1 - (kk - 1)) * HY + (ty + 1 - (jj - 1));
                    const bool full = CONN == 26 || (CONN == 18 && nz <= 1) || (CONN == 6 && nz == 0); // x pattern 111
                    const bool mid = (CONN == 18 && nz == 2) || (CONN == 6 && nz == 1);                // x pattern 010
                    v[kk * 3 + jj] = full ? L.sD[src] : (mid ? L.sN[src] : 0ull);
                }
#pragma unroll
            for (int q = 0; q < 9; q++) nb |= v[q];
        } else {
#pragma unroll
        for (int kk = 0; kk < 3; kk++)
#pragma unroll
            for (int jj = 0; jj < 3; jj++) {
                const uint32_t m3 = (st >> (kk * 9 + jj * 3)) & 7u;
                if (!m3) continue; // uniform
                // voxel p reached => p + (kk-1, jj-1, ii-1) reached; gather: source row = (z-(kk-1), y-(jj-1))
                const int src = (tz + 1 - (kk - 1)) * HY + (ty + 1 - (jj - 1));
                if (m3 == 7u) nb |= L.sD[src];
                else {
                    const unsigned long long n = L.sN[src];
                    if (m3 & 2u) nb |= n;
                    if (m3 & 4u) nb |= (n << 1) | (unsigned long long)L.sCL[src];        // ii = 2: source bit x-1
                    if (m3 & 1u) nb |= (n >> 1) | ((unsigned long long)L.sCR[src] << 63); // ii = 0: source bit x+1
                }
            }
        }
        unsigned long long nr = r | (nb & c);
        if (xrun) nr = fill_runs(nr, c);
        const bool ch = nr != r;
        if (ch) {
            r = nr;
            L.sN[me] = r;
            L.sD[me] = r | (r << 1) | (r >> 1) | mycl | (mycr << 63);
        }
        changed |= ch;
      }
        // vote: __syncthreads_or() costs ~1500 cycles on gfx950 (library workgroup reduction); a wave ballot + one
        // LDS flag + one barrier does the same in ~200.  Flag it&1 is read now, flag (it+1)&1 is cleared for the next
        // iteration (nobody touches it between this barrier and the next one's writers).
        // (Tried: one flag per wave, a wave whose own and whose neighbour waves' slices stood still in the last iteration
        // skips its gather -- 0.192 -> 0.213 ms on the bench volume: the fronts move along y as much as along z, so all
        // four waves stay busy, and the flag read at the loop head is one more dependent LDS round trip.)
        if (__any(changed) && (threadIdx.x & 63) == 0) L.vote[it & 1] = 1u;
        if (threadIdx.x == 0) L.vote[(it + 1) & 1] = 0u;
        __syncthreads();
        if (!L.vote[it & 1]) {
            exhausted = false;
            break;
        }
    }
    DBG_T(2)
    if (DBG && threadIdx.x == 0) g_dbg[8] = 0;
    const unsigned long long chg = r ^ r_in;
    unsigned dirs = 0;
    if (chg) {
        if (ATOMIC) atomicOr(&reached[(z * t.dy + y) * t.wx + txi], r); // monotone publish: bits are never lost
        else reached[(z * t.dy + y) * t.wx + txi] = r;
        // which neighbour tiles can see this change?  direction d = (dz+1)*9 + (dy+1)*3 + (dx+1)
        const bool zlo = tz == 0, zhi = tz == TZ - 1, ylo = ty == 0, yhi = ty == TY - 1;
        const bool xlo = chg & 1ull, xhi = chg >> 63;
#pragma unroll
        for (int dzz = -1; dzz <= 1; dzz++)
#pragma unroll
            for (int dyy = -1; dyy <= 1; dyy++)
#pragma unroll
                for (int dxx = -1; dxx <= 1; dxx++) {
                    if (!dzz && !dyy && !dxx) continue;
                    const bool vis = (dzz == 0 || (dzz < 0 ? zlo : zhi)) && (dyy == 0 || (dyy < 0 ? ylo : yhi)) &&
                                     (dxx == 0 || (dxx < 0 ? xlo : xhi));
                    dirs |= vis ? (1u << ((dzz + 1) * 9 + (dyy + 1) * 3 + (dxx + 1))) : 0u;
                }
        if (exhausted) dirs |= 1u << 13; // iteration cap hit before the local fix-point: revisit this tile
    }
    // OR across the wave in registers, then ONE LDS atomic per wave (256 same-address LDS atomics serialise)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dirs |= __shfl_xor(dirs, o, 64);
    if ((threadIdx.x & 63) == 0 && dirs) atomicOr(&L.dirs, dirs);
    if (__any((c & ~r) != 0ull) && (threadIdx.x & 63) == 0) L.open = 1u;
}

// ---- directed tile update (floodfill_auto_threshold) -------------------------------------------------------------
// The admissible range of a step depends on the voxel it LEAVES (floodfill_py.rs:32-35), so reachability is directed:
// six edge planes say, per source voxel, whether the step to its +x / -x / +y / -y / +z / -z neighbour is allowed
// (k_flood_edges_auto folds "the target is not a barrier" into them).  Same tile, same relaxation, same votes as
// tile_update; a lane gathers  reached(neighbour row) & edge(neighbour row -> me)  for its four row neighbours (the edge
// words are loaded once per visit), and closes x-chains inside its word with the carry trick along the edge bits:
// (E + (R & E)) ^ E  sets every bit a run of consecutive edges carries a reached bit to (bit-reversed twin for -x).
__device__ __forceinline__ void tile_update_dir(const Tiles &t, const unsigned long long *__restrict__ edges,
                                                unsigned long long *reached, int64_t tile, TileLds &L) {
    const int64_t txi = tile % t.wx, r1 = tile / t.wx;
    const int64_t tyi = r1 % t.nty, tzi = r1 / t.nty;
    const int64_t z0 = tzi * TZ, y0 = tyi * TY;
    const int64_t plane = t.dz * t.dy * t.wx; // words per edge plane
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int idx = threadIdx.x + pass * NT;
        if (idx < HZ * HY) {
            const int yy = idx % HY, zz = idx / HY;
            const int64_t z = z0 + zz - 1, y = y0 + yy - 1;
            L.sN[idx] = (z >= 0 && z < t.dz && y >= 0 && y < t.dy) ? reached[(z * t.dy + y) * t.wx + txi] : 0ull;
        }
    }
    const int ty = threadIdx.x & (TY - 1), tz = threadIdx.x >> TY_LOG;
    const int64_t z = z0 + tz, y = y0 + ty;
    const bool inside = z < t.dz && y < t.dy;
    const int64_t me_w = (z * t.dy + y) * t.wx + txi;
    unsigned long long exp = 0, bexm = 0, e_ym = 0, e_yp = 0, e_zm = 0, e_zp = 0, carry = 0;
    if (inside) {
        exp = edges[me_w];
        bexm = __brevll(edges[plane + me_w]);
        if (y > 0) e_ym = edges[2 * plane + me_w - t.wx];                  // row y-1 stepping +y
        if (y + 1 < t.dy) e_yp = edges[3 * plane + me_w + t.wx];           // row y+1 stepping -y
        if (z > 0) e_zm = edges[4 * plane + me_w - t.dy * t.wx];           // slice z-1 stepping +z
        if (z + 1 < t.dz) e_zp = edges[5 * plane + me_w + t.dy * t.wx];    // slice z+1 stepping -z
        // steps across the word boundary come from the x-neighbour tiles: constant during this visit
        if (txi > 0) carry |= (reached[me_w - 1] & edges[me_w - 1]) >> 63;
        if (txi + 1 < t.wx) carry |= (reached[me_w + 1] & edges[plane + me_w + 1] & 1ull) << 63;
    }
    if (threadIdx.x == 0) L.vote[0] = 0u;
    __syncthreads();
    const int me = (tz + 1) * HY + (ty + 1);
    const unsigned long long r_in = L.sN[me];
    unsigned long long r = r_in;
    bool exhausted = true;
    for (int it = 0; it < t.itcap; it++) {
        unsigned long long nr = r | carry | (L.sN[me - 1] & e_ym) | (L.sN[me + 1] & e_yp) | (L.sN[me - HY] & e_zm) |
                                (L.sN[me + HY] & e_zp);
        nr |= (exp + (nr & exp)) ^ exp;
        unsigned long long rn = __brevll(nr);
        rn |= (bexm + (rn & bexm)) ^ bexm;
        nr = __brevll(rn);
        if (!inside) nr = 0ull;
        const bool ch = nr != r;
        if (ch) {
            r = nr;
            L.sN[me] = r;
        }
        if (__any(ch) && (threadIdx.x & 63) == 0) L.vote[it & 1] = 1u;
        if (threadIdx.x == 0) L.vote[(it + 1) & 1] = 0u;
        __syncthreads();
        if (!L.vote[it & 1]) {
            exhausted = false;
            break;
        }
    }
    const unsigned long long chg = r ^ r_in;
    unsigned dirs = 0;
    if (chg) {
        reached[me_w] = r;
        const bool zlo = tz == 0, zhi = tz == TZ - 1, ylo = ty == 0, yhi = ty == TY - 1;
        const bool xlo = chg & 1ull, xhi = chg >> 63;
        // face neighbours only: every step moves along one axis
        if (zlo) dirs |= 1u << 4;
        if (zhi) dirs |= 1u << 22;
        if (ylo) dirs |= 1u << 10;
        if (yhi) dirs |= 1u << 16;
        if (xlo) dirs |= 1u << 12;
        if (xhi) dirs |= 1u << 14;
        if (exhausted) dirs |= 1u << 13;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dirs |= __shfl_xor(dirs, o, 64);
    if ((threadIdx.x & 63) == 0 && dirs) atomicOr(&L.dirs, dirs);
    if (threadIdx.x == 0) L.open = 1u; // (edge planes: no candidate plane to be exhausted)
}

// ---- linear-index tile update (the IFT watershed's cost levels, ivx_dev_ws_cost_levels) --------------------------------
// scipy's watershed_ift takes neighbours by LINEAR index (ni_measure.c: only indices outside [0, size) are refused), so the
// last voxel of a row neighbours the first voxel of the next row, and the last row of a slice the first row of the next
// slice.  With rows of whole 64-voxel words the volume is ONE flat bit array: the x-neighbour of a word is word +- 1, the
// y-neighbour word +- wx, the z-neighbour word +- dy * wx, and "in bounds" means 0 <= word < nwords -- nothing else.  Arcs are
// symmetric, so three planes suffice: bit b of Ex / Ey / Ez[word] = the arc from voxel (word, b) to its +1 / +W / +HW
// neighbour is enabled; the arc arriving from the other side is the neighbour word's bit.  The staged halo rows are the
// LINEAR neighbours (row index z * dy + y, taken as it comes), which is the wrap-around for free; only the wake-up of the
// tiles that read a changed word has to look the reader up when it is not the lattice neighbour (border tiles).
// (returns true when the reader is this very tile -- a volume one tile wide: the caller then revisits itself)
__device__ __forceinline__ bool lin_enlist(const Tiles &t, int64_t word, int64_t self, uint8_t *dirty_next, unsigned int *list_next,
                                           unsigned int *n_next) {
    const int64_t row = word / t.wx, txi = word - row * t.wx, z = row / t.dy, y = row - z * t.dy;
    const int64_t nt = ((z / TZ) * t.nty + (y >> TY_LOG)) * t.wx + txi;
    if (nt == self) return true;
    unsigned int *wp = (unsigned int *)(dirty_next + (nt & ~(int64_t)3));
    const unsigned int sh = 8 * (unsigned int)(nt & 3);
    const unsigned int old = atomicOr(wp, 1u << sh);
    if (!((old >> sh) & 0xffu)) list_next[atomicAdd(n_next, 1u)] = (unsigned int)nt;
    return false;
}

__device__ __forceinline__ void tile_update_lin(const Tiles &t, const unsigned long long *__restrict__ E,
                                                unsigned long long *reached, int64_t tile, TileLds &L, uint8_t *dirty_next,
                                                unsigned int *list_next, unsigned int *n_next) {
    const int64_t txi = tile % t.wx, r1 = tile / t.wx;
    const int64_t tyi = r1 % t.nty, tzi = r1 / t.nty;
    const int64_t z0 = tzi * TZ, y0 = tyi * TY;
    const int64_t nrows = t.dz * t.dy, nwords = nrows * t.wx, hwx = t.dy * t.wx;
    const unsigned long long *Ex = E, *Ey = E + nwords, *Ez = E + 2 * nwords;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int idx = threadIdx.x + pass * NT;
        if (idx < HZ * HY) {
            const int yy = idx % HY, zz = idx / HY;
            const int64_t row = (z0 + zz - 1) * t.dy + (y0 + yy - 1); // linear: row -1 of a slice IS the last row of the slice below
            L.sN[idx] = (row >= 0 && row < nrows) ? reached[row * t.wx + txi] : 0ull;
        }
    }
    const int ty = threadIdx.x & (TY - 1), tz = threadIdx.x >> TY_LOG;
    const int64_t row = (z0 + tz) * t.dy + (y0 + ty);
    const bool inside = row < nrows; // (dy is a multiple of TY: only whole slices can be missing)
    const int64_t me_w = row * t.wx + txi;
    unsigned long long exp = 0, bexm = 0, e_ym = 0, e_yp = 0, e_zm = 0, e_zp = 0, carry = 0, xgain = 0;
    if (inside) {
        exp = Ex[me_w];
        bexm = __brevll(exp << 1); // stepping down from bit b needs the arc (b - 1, b) = bit b - 1 of Ex
        if (me_w - t.wx >= 0) e_ym = Ey[me_w - t.wx];
        if (me_w + t.wx < nwords) e_yp = Ey[me_w];
        if (me_w - hwx >= 0) e_zm = Ez[me_w - hwx];
        if (me_w + hwx < nwords) e_zp = Ez[me_w];
        // steps across the word boundary come from the neighbour WORDS (other tiles): constant during this visit
        if (me_w > 0) {
            const unsigned long long lr = reached[me_w - 1] >> 63, le = Ex[me_w - 1] >> 63;
            carry |= lr & le;
            xgain |= le & ~lr;              // my bit 0 could still hand something to the left word
        }
        if (me_w + 1 < nwords) {
            const unsigned long long rr = reached[me_w + 1] & 1ull, re = exp >> 63;
            carry |= (rr & re) << 63;
            xgain |= (re & ~rr) << 63;      // my bit 63 to the right word
        }
    }
    if (threadIdx.x == 0) L.vote[0] = 0u;
    __syncthreads();
    const int me = (tz + 1) * HY + (ty + 1);
    const unsigned long long r_in = L.sN[me];
    unsigned long long r = r_in;
    bool exhausted = true;
    for (int it = 0; it < t.itcap; it++) {
        unsigned long long nr = r | carry | (L.sN[me - 1] & e_ym) | (L.sN[me + 1] & e_yp) | (L.sN[me - HY] & e_zm) |
                                (L.sN[me + HY] & e_zp);
        nr |= (exp + (nr & exp)) ^ exp;
        unsigned long long rn = __brevll(nr);
        rn |= (bexm + (rn & bexm)) ^ bexm;
        nr = __brevll(rn);
        if (!inside) nr = 0ull;
        const bool ch = nr != r;
        if (ch) {
            r = nr;
            L.sN[me] = r;
        }
        if (__any(ch) && (threadIdx.x & 63) == 0) L.vote[it & 1] = 1u;
        if (threadIdx.x == 0) L.vote[(it + 1) & 1] = 0u;
        __syncthreads();
        if (!L.vote[it & 1]) {
            exhausted = false;
            break;
        }
    }
    const unsigned long long chg = r ^ r_in;
    unsigned dirs = 0;
    if (chg) {
        reached[me_w] = r;
        // A neighbour tile is woken only when one of the NEW bits has an enabled arc to a voxel of it that was not
        // reached when this visit staged its halo (a stale halo can only wake one tile too many): in a flood near the
        // percolation threshold most changes have nowhere to go, and most wake-ups were visits that found nothing.
        const bool zlo = tz == 0 && (chg & e_zm & ~L.sN[me - HY]), zhi = tz == TZ - 1 && (chg & e_zp & ~L.sN[me + HY]);
        const bool ylo = ty == 0 && (chg & e_ym & ~L.sN[me - 1]), yhi = ty == TY - 1 && (chg & e_yp & ~L.sN[me + 1]);
        const bool xlo = chg & xgain & 1ull, xhi = (chg & xgain) >> 63;
        // readers that ARE the lattice neighbour tile: one direction bit per workgroup (enlisted by the caller);
        // readers across a row / slice end (border tiles only): looked up word by word
        if (zlo) dirs |= 1u << 4;
        if (zhi) dirs |= 1u << 22;
        if (ylo) {
            if (y0 > 0) dirs |= 1u << 10;
            else if (me_w - t.wx >= 0 && lin_enlist(t, me_w - t.wx, tile, dirty_next, list_next, n_next)) dirs |= 1u << 13;
        }
        if (yhi) {
            if (y0 + TY < t.dy) dirs |= 1u << 16;
            else if (me_w + t.wx < nwords && lin_enlist(t, me_w + t.wx, tile, dirty_next, list_next, n_next)) dirs |= 1u << 13;
        }
        if (xlo) {
            if (txi > 0) dirs |= 1u << 12;
            else if (me_w > 0 && lin_enlist(t, me_w - 1, tile, dirty_next, list_next, n_next)) dirs |= 1u << 13;
        }
        if (xhi) {
            if (txi + 1 < t.wx) dirs |= 1u << 14;
            else if (me_w + 1 < nwords && lin_enlist(t, me_w + 1, tile, dirty_next, list_next, n_next)) dirs |= 1u << 13;
        }
        if (exhausted) dirs |= 1u << 13;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dirs |= __shfl_xor(dirs, o, 64);
    if ((threadIdx.x & 63) == 0 && dirs) atomicOr(&L.dirs, dirs);
    if (threadIdx.x == 0) L.open = 1u; // (arc planes change from level to level: a tile is never closed for good)
}

// diagnostic only (not part of include/ivx.h): cycle stamps written under IVX_FLOOD_DBG=1, see tools/dbg_tile.py
extern "C" int ivx_debug_read(unsigned long long *out16) {
    IVX_HIP(hipDeviceSynchronize());
    IVX_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dbg), 16 * 8));
    return IVX_OK;
}

// ---- list-driven round: only as many workgroups as there are dirty tiles do any work ------------------------------------
// The dirty set of a round is a compact list (plus the byte flags, which de-duplicate marks).  A workgroup takes list
// entries blockIdx.x, blockIdx.x + gridDim.x, ...; an idle round costs one small launch instead of ntiles empty workgroups,
// and the dirty tiles of a busy round start at once instead of waiting for the dispatcher to walk past the clean ones.
__global__ void k_flood_build_list(Tiles t, const uint8_t *__restrict__ dirty, unsigned int *__restrict__ list,
                                   unsigned int *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t.ntiles && (dirty[i] & 1u)) list[atomicAdd(count, 1u)] = (unsigned int)i;
}

// `line` (pinned host memory, ivx::progress_line): word 0 = tag | round + 1 | tiles on this round's list, stored as the
// round STARTS (everything before it in the stream is complete, so the count is final); word 1 = tag | round + 1 of the
// last round that had any work.  The host polls word 0, keeps a few rounds queued ahead of the newest one it has seen
// start, and stops when a round starts with an empty list.
// MODE 0: candidate plane; 1 (directed): `cand` holds the six edge planes of k_flood_edges_auto instead; 2 (linear): the
// three arc planes of k_wsa_edges and scipy's linear-index neighbourhood (tile_update_lin)
template <int MODE>
__global__ __launch_bounds__(NT, 6) void k_flood_round_list(Tiles t, const unsigned long long *__restrict__ cand,
                                                             unsigned long long *reached, const unsigned int *__restrict__ list_cur,
                                                             const unsigned int *__restrict__ n_cur, uint8_t *dirty_cur,
                                                             uint8_t *dirty_next, unsigned int *list_next,
                                                             unsigned int *n_next, unsigned int *n_clear,
                                                             unsigned long long *line, unsigned int tag_round,
                                                             unsigned int *gate, unsigned int gate_val,
                                                             unsigned int gate_below) {
    __shared__ TileLds L;
    const unsigned int n = *n_cur;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // a short list: the busy rounds are over, let background work gated on this word start (ivx_dev_flood_arm_gate)
        if (gate && n < gate_below) __hip_atomic_store(gate, gate_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *n_clear = 0u; // the counter the round AFTER the next one appends to
        const unsigned long long hi = (unsigned long long)tag_round << 32;
        __hip_atomic_store(&line[0], hi | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (n) __hip_atomic_store(&line[1], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (unsigned int li = blockIdx.x; li < n; li += gridDim.x) {
        const int64_t tile = list_cur[li];
        if (threadIdx.x == 0) {
            dirty_cur[tile] = 0;
            L.dirs = 0;
            L.open = 0;
        }
        __syncthreads();
        const bool dbg = (t.strct >> 30 & 1u) && li == 0; // IVX_FLOOD_DBG: cycle stamps of the first tile (tools/dbg_tile.py)
        if (dbg && threadIdx.x == 0) g_dbg[0] = __builtin_readcyclecounter();
        if (MODE == 2) tile_update_lin(t, cand, reached, tile, L, dirty_next, list_next, n_next);
        else if (MODE == 1) tile_update_dir(t, cand, reached, tile, L);
        else if (dbg) tile_update<false, 26, true>(t, cand, reached, tile, L);
        else if (t.conn == 26) tile_update<false, 26>(t, cand, reached, tile, L);
        else if (t.conn == 18) tile_update<false, 18>(t, cand, reached, tile, L);
        else if (t.conn == 6) tile_update<false, 6>(t, cand, reached, tile, L);
        else tile_update<false, 0>(t, cand, reached, tile, L);
        lds_barrier(); // L.dirs complete; the publish stores keep flying (the kernel boundary orders them for the next round)
        if (dbg && threadIdx.x == 0) g_dbg[3] = __builtin_readcyclecounter();
        const int64_t txi = tile % t.wx, r1 = tile / t.wx;
        const int64_t tyi = r1 % t.nty, tzi = r1 / t.nty;
        if (threadIdx.x == 32 && !L.open) { // closed for good: nobody needs to enlist this tile again (see CLOSED)
            const unsigned int bit = CLOSED << (8 * (unsigned int)(tile & 3));
            atomicOr((unsigned int *)(dirty_cur + (tile & ~(int64_t)3)), bit);
            atomicOr((unsigned int *)(dirty_next + (tile & ~(int64_t)3)), bit);
        }
        if (threadIdx.x < 27 && (L.dirs >> threadIdx.x & 1u)) {
            const int d = threadIdx.x;
            const int64_t nz = tzi + d / 9 - 1, ny = tyi + (d / 3) % 3 - 1, nx = txi + d % 3 - 1;
            if (nz >= 0 && nz < t.ntz && ny >= 0 && ny < t.nty && nx >= 0 && nx < t.wx) {
                const int64_t nt = (nz * t.nty + ny) * t.wx + nx;
                unsigned int *wp = (unsigned int *)(dirty_next + (nt & ~(int64_t)3));
                const unsigned int sh = 8 * (unsigned int)(nt & 3);
                const unsigned int old = atomicOr(wp, 1u << sh);
                if (!((old >> sh) & 0xffu)) list_next[atomicAdd(n_next, 1u)] = (unsigned int)nt; // first mark: enlist
            }
        }
        if (dbg && threadIdx.x == 0) g_dbg[4] = __builtin_readcyclecounter();
        __syncthreads(); // L is reused by the next list entry of this workgroup
    }
}

#define AT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// ---- resident rounds: the same rounds, ONE launch ------------------------------------------------------------------
// A launch per round costs ~5 us of dispatch whether the round has 1 500 tiles or none, the host has to see a round start
// with an empty list before it stops queueing (three more empty launches are in the queue by then), and the stage that
// follows waits behind all of them.  Here the grid stays resident (every workgroup is on a compute unit at once: the host
// sizes the grid from the occupancy) and a round boundary is a device-wide barrier: a workgroup drains its own stores and
// atomics (s_waitcnt), signs an arrival counter and polls it.  Everything that crosses workgroups inside the launch --
// reached words, dirty bytes, list entries, counters -- is an agent-scope access (served where all XCDs meet), so the
// boundary needs no L2 write-back.  Only workgroups that had a tile this round sign (the others just wait for them).
// The kernel ends at the first empty list, or after `round_cap` rounds (the caller's escape to the union-find engine), or
// -- every spin is bounded -- with status 3 when a barrier never completes (the caller finishes with launches per round:
// `reached` is monotone, a partial result is valid).  The last word goes to the pinned progress line.
// OPT-IN (IVX_FLOOD_RESIDENT=1), measured slower on the bench volume: region growing 0.27 ms with one workgroup per compute
// unit, 0.31 / 0.42 / 0.49 ms with 2 / 4 / 6, against 0.19 ms for one launch per round -- a round inside the launch is a chain
// of ~6 agent-scope round trips of ~1.2 us (poll, list entry, staged rows, publish, drain, arrival) and up to 10^3 pollers
// share one address, while a kernel boundary costs ~5 us flat and its loads hit the L2.  What the launch per round cannot
// do -- have the next stage queued before the flood ends -- ivx_dev_flood_grow_async / ivx_dev_flood_wait offer with it.
struct ResCtl {
    unsigned int arrive; // barrier arrivals since the launch (monotone)
    unsigned int status; // 0 running, 1 converged, 2 round cap reached, 3 a barrier timed out
    unsigned int rounds; // rounds that had work
    unsigned int visits; // (statistics) tile visits
};
constexpr int RES_CTL_AT = 32; // dword index of ResCtl inside the counter block (cleared together with the ring)
__global__ __launch_bounds__(NT, 6) void k_flood_resident(Tiles t, const unsigned long long *__restrict__ cand,
                                                           unsigned long long *reached, unsigned int *list0,
                                                           unsigned int *list1, unsigned int *cnt, uint8_t *dirty0,
                                                           uint8_t *dirty1, ResCtl *ctl, unsigned long long *line,
                                                           unsigned int tag, unsigned int ring, unsigned int round_cap,
                                                           unsigned int max_spins, unsigned int *gate, unsigned int gate_val,
                                                           unsigned int gate_below) {
    __shared__ TileLds L;
    __shared__ unsigned int s_n, s_abort;
    unsigned int target = 0, round = 0, status = 1;
    for (;; round++) {
        const unsigned int r = round % ring, cur = round & 1u;
        if (threadIdx.x == 0) {
            s_n = AT_LOAD(&cnt[r]);
            s_abort = 0u;
        }
        __syncthreads();
        const unsigned int n = s_n;
        if (n == 0u) break;
        if (round >= round_cap) {
            status = 2;
            break;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            AT_STORE(&cnt[(r + 2u) % ring], 0u); // the counter the round AFTER the next one appends to
            if (gate && n < gate_below) AT_STORE(gate, gate_val);
        }
        unsigned int *list_cur = cur ? list1 : list0, *list_next = cur ? list0 : list1;
        uint8_t *dirty_cur = cur ? dirty1 : dirty0, *dirty_next = cur ? dirty0 : dirty1;
        unsigned int *n_next = cnt + (r + 1u) % ring;
        for (unsigned int li = blockIdx.x; li < n; li += gridDim.x) {
            const int64_t tile = AT_LOAD(&list_cur[li]);
            if (threadIdx.x == 0) {
                __hip_atomic_store(&dirty_cur[tile], (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                L.dirs = 0;
                L.open = 0;
            }
            __syncthreads();
            if (t.conn == 26) tile_update<true, 26>(t, cand, reached, tile, L);
            else if (t.conn == 18) tile_update<true, 18>(t, cand, reached, tile, L);
            else if (t.conn == 6) tile_update<true, 6>(t, cand, reached, tile, L);
            else tile_update<true, 0>(t, cand, reached, tile, L);
            lds_barrier(); // L.dirs complete; the publishing atomics keep flying until the round's barrier
            const int64_t txi = tile % t.wx, r1 = tile / t.wx;
            const int64_t tyi = r1 % t.nty, tzi = r1 / t.nty;
            if (threadIdx.x == 32 && !L.open) { // closed for good (see CLOSED)
                const unsigned int bit = CLOSED << (8 * (unsigned int)(tile & 3));
                atomicOr((unsigned int *)(dirty_cur + (tile & ~(int64_t)3)), bit);
                atomicOr((unsigned int *)(dirty_next + (tile & ~(int64_t)3)), bit);
            }
            if (threadIdx.x < 27 && (L.dirs >> threadIdx.x & 1u)) {
                const int d = threadIdx.x;
                const int64_t nz = tzi + d / 9 - 1, ny = tyi + (d / 3) % 3 - 1, nx = txi + d % 3 - 1;
                if (nz >= 0 && nz < t.ntz && ny >= 0 && ny < t.nty && nx >= 0 && nx < t.wx) {
                    const int64_t nt = (nz * t.nty + ny) * t.wx + nx;
                    unsigned int *wp = (unsigned int *)(dirty_next + (nt & ~(int64_t)3));
                    const unsigned int sh = 8 * (unsigned int)(nt & 3);
                    const unsigned int old = atomicOr(wp, 1u << sh);
                    if (!((old >> sh) & 0xffu)) AT_STORE(&list_next[atomicAdd(n_next, 1u)], (unsigned int)nt);
                }
            }
            __syncthreads(); // L is reused by the next list entry of this workgroup
        }
        // the round's barrier: min(n, grid) workgroups had work and sign; everybody waits for all of them
        target += n < gridDim.x ? n : gridDim.x;
        __builtin_amdgcn_s_waitcnt(0); // this wave's stores and atomics have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) {
            if (blockIdx.x < n) atomicAdd(&ctl->arrive, 1u);
            for (unsigned int spins = 0; (int)(AT_LOAD(&ctl->arrive) - target) < 0; spins++) {
                if (spins > max_spins || AT_LOAD(&ctl->status) == 3u) {
                    AT_STORE(&ctl->status, 3u);
                    s_abort = 1u;
                    break;
                }
                if (blockIdx.x < n) __builtin_amdgcn_s_sleep(2);
                else __builtin_amdgcn_s_sleep(24); // (a workgroup without work this round: poll gently)
            }
        }
        __syncthreads();
        if (s_abort) {
            status = 3;
            break;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (status != 3u) AT_STORE(&ctl->status, status);
        AT_STORE(&ctl->rounds, round);
        if (gate) AT_STORE(gate, gate_val); // whatever happened, the gate is open when the flood is over
        __hip_atomic_store(&line[0], ((unsigned long long)tag << 56) | ((unsigned long long)status << 48) | round, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- persistent tile frontier: ONE launch, device-side work queue ---------------------------------------
// Same tile update as k_flood_round_list, but workgroups pull dirty tiles from a ring buffer and push the neighbour
// tiles whose halo they changed, until nothing is queued or in flight (`pending` == 0).  Cross-workgroup traffic
// (reached words, queue words, flags) uses 4/8-byte agent-scope atomics on both sides (the placement-independent
// form of the CDNA4 guide, G16): words are published with atomicOr (monotone, so two workgroups that happen to
// own the same tile concurrently can never lose bits), drained with s_waitcnt vmcnt(0) before the push.
// No co-residency is needed: a workgroup only ever waits for work while some OTHER running workgroup holds a tile.

__global__ void k_flood_enqueue(Tiles t, uint8_t *dirty, Queue *q, unsigned int *queued, unsigned int *ring,
                                unsigned int qmask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.ntiles || !(dirty[i] & 1u)) return;
    dirty[i] = 0;
    if (atomicExch(&queued[i], 1u) == 0u) {
        atomicAdd(&q->pending, 1u);
        const unsigned int slot = atomicAdd(&q->tail, 1u);
        AT_STORE(&ring[slot & qmask], (unsigned int)i);
    }
}

__global__ __launch_bounds__(NT) void k_flood_persistent(Tiles t, const unsigned long long *__restrict__ cand,
                                                          unsigned long long *reached, Queue *q, unsigned int *queued,
                                                          unsigned int *ring, unsigned int qmask, unsigned int max_spins) {
    __shared__ TileLds L;
    __shared__ int s_tile;
    unsigned int my_visits = 0;
    // nothing was enqueued (the counter is final: the enqueue kernel has completed): nobody would ever set `done`
    if (blockIdx.x == 0 && threadIdx.x == 0 && AT_LOAD(&q->pending) == 0u) AT_STORE(&q->done, 1u);
    for (;;) {
        if (threadIdx.x == 0) {
            int tile = -1;
            if (!AT_LOAD(&q->done) && !AT_LOAD(&q->abort)) {
                // take a ticket: the h-th pop gets the h-th push; then poll ONLY our own ring slot (no hot word)
                const unsigned int h = atomicAdd(&q->head, 1u);
                unsigned int *slot = &ring[h & qmask];
                for (unsigned int spins = 0;; spins++) {
                    const unsigned int v = AT_LOAD(slot);
                    if (v != Q_EMPTY) {
                        AT_STORE(slot, Q_EMPTY);
                        tile = (int)v;
                        break;
                    }
                    if ((spins & 7u) == 7u && (AT_LOAD(&q->done) || AT_LOAD(&q->abort))) break;
                    if (spins > max_spins) { AT_STORE(&q->abort, 1u); break; } // bounded: never hang the device
                    if (spins < 64u) __builtin_amdgcn_s_sleep(2);
                    else __builtin_amdgcn_s_sleep(32);
                }
            }
            if (tile >= 0 && ++my_visits > max_spins) { AT_STORE(&q->abort, 2u); tile = -1; }
            if (tile >= 0) atomicExch(&queued[tile], 0u); // cleared BEFORE staging: later changes re-queue the tile
            s_tile = tile;
            L.dirs = 0;
            L.open = 0;
        }
        __syncthreads();
        const int tile = s_tile;
        if (tile < 0) return;
        if (t.conn == 26) tile_update<true, 26>(t, cand, reached, tile, L);
        else if (t.conn == 18) tile_update<true, 18>(t, cand, reached, tile, L);
        else if (t.conn == 6) tile_update<true, 6>(t, cand, reached, tile, L);
        else tile_update<true, 0>(t, cand, reached, tile, L);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every publishing wave drains before the pushes
        __syncthreads();
        const int64_t txi = tile % t.wx, r1 = tile / t.wx;
        const int64_t tyi = r1 % t.nty, tzi = r1 / t.nty;
        if (threadIdx.x < 27 && (L.dirs >> threadIdx.x & 1u)) {
            const int d = threadIdx.x;
            const int64_t nz = tzi + d / 9 - 1, ny = tyi + (d / 3) % 3 - 1, nx = txi + d % 3 - 1;
            if (nz >= 0 && nz < t.ntz && ny >= 0 && ny < t.nty && nx >= 0 && nx < t.wx) {
                const int64_t nt = (nz * t.nty + ny) * t.wx + nx;
                if (atomicExch(&queued[nt], 1u) == 0u) {
                    atomicAdd(&q->pending, 1u); // counted before it becomes poppable
                    const unsigned int slot = atomicAdd(&q->tail, 1u);
                    AT_STORE(&ring[slot & qmask], (unsigned int)nt);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // all pushes (and their pending increments) have landed before this tile is retired
        if (threadIdx.x == 0) {
            atomicAdd(&q->visits, 1u);
            if (atomicSub(&q->pending, 1u) == 1u) AT_STORE(&q->done, 1u); // nothing queued, nothing in flight
        }
    }
}

__global__ void k_flood_mark(Tiles t, int64_t tz0, int64_t tz1, uint8_t *dirty) {
    const int64_t per = t.nty * t.wx;
    const int64_t n = (tz1 - tz0) * per;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dirty[tz0 * per + i] = 1;
}

// ---- coarse pass: whole tiles at once ---------------------------------------------------------------
// A tile whose in-bounds voxels are ALL candidates is connected in itself under every structuring element that
// contains the six face offsets, so one reached voxel in it means the whole tile is reached, and two such tiles that
// touch (by a face; by an edge for 18/26; by a corner for 26) reach each other.  The interior of a solid region is made
// of such tiles, and crossing it voxel-tile by voxel-tile costs one ~11 us round per tile hop.  The coarse pass floods
// the TILE graph instead: flags (k_flood_tile_flags) -> fix-point over rows of tile bits in ONE workgroup
// (k_flood_coarse: a row of tiles along x is one 64-bit word, the same dilate / carry-fill update as the voxel tiles)
// -> reached = cand for the tiles it reached and a dirty mark on their other neighbours (k_flood_coarse_apply).
// The ordinary rounds then only have the boundary shell left.  Result bits are identical with or without it.
constexpr int CT = 1024, CRP = 8, CROWS_MAX = 7680; // coarse workgroup size, rows per lane, LDS rows incl. halo (60 KB)
constexpr unsigned int ENLISTED = 0x80u;             // dirty-byte bit: "already on the round-0 list" (coarse pass only)

// Granularity of the coarse graph: blocks of BXS x 16 x 16 voxels, BXS = 16, 32 or 64 (a quarter, half or whole tile along
// x; the y / z extent is the tile's, so a row of blocks along x is still one row of the tile grid).  A row of blocks is one
// 64-bit word (bit = block index along x), so BXS is the smallest of the three with 64 / BXS * wx <= 64: 16 up to 1024
// voxels along x, 32 up to 2048, 64 (= the tile itself) up to 4096.  Finer blocks reach closer to the region's surface:
// on the bench volume the rounds that follow drop from 13 to 8 (tools/sim_flood.c reproduces the round structure on the CPU).
//   rowF[g]  the block is all-candidate,   rowW[g]  the block is wholly reached (seeded here / by k_flood_coarse, closed by it).
struct Blocks {
    int q;   // blocks per tile along x (4, 2, 1)
    int bxs; // voxels per block along x (16, 32, 64)
};
__device__ __forceinline__ unsigned long long block_mask(const Blocks &b, unsigned long long qbits) {
    // qbits: one bit per block of a word -> the voxel bits of those blocks
    const unsigned long long one = b.bxs == 64 ? ~0ull : ((1ull << b.bxs) - 1ull);
    unsigned long long m = 0;
    for (int q = 0; q < b.q; q++)
        if (qbits >> q & 1ull) m |= one << (b.bxs * q);
    return m;
}
// the blocks of tile column txi that exist (hold at least one in-bounds voxel), as a q-bit mask
__device__ __forceinline__ unsigned int block_exist(const Tiles &t, const Blocks &b, int64_t txi) {
    unsigned int m = 0;
    for (int q = 0; q < b.q; q++)
        if (txi * 64 + (int64_t)b.bxs * q < t.dx) m |= 1u << q;
    return m;
}

__device__ __forceinline__ double load_as_double(const void *data, int dtype, int64_t i) {
    switch (dtype) {
    case IVX_I16: return (double)((const int16_t *)data)[i];
    case IVX_U8: return (double)((const uint8_t *)data)[i];
    case IVX_U16: return (double)((const uint16_t *)data)[i];
    default: return ((const double *)data)[i];
    }
}

// One workgroup per tile row: the 16 rows of a slice are contiguous, so the loads coalesce across the row's tiles.
// FRESH: the flood starts from the seeds in `sp` and nothing else -- `reached` and the dirty flags are not read (their old
// contents are dead: k_flood_block_apply<true> rewrites every word of the plane), this kernel zeroes the dirty flags and
// the round counters, and workgroup 0 tests the seeds (floodfill.rs:123: only in-range seeds start a flood; an accepted
// seed is a candidate even when its barrier byte already equals `fill`) and leaves the accepted ones as a bit mask.
template <bool FRESH>
__global__ __launch_bounds__(256) void k_flood_block_flags(Tiles t, Blocks bk, unsigned long long *cand,
                                                           const unsigned long long *__restrict__ reached,
                                                           uint8_t *dirty, uint8_t *dirty1,
                                                           unsigned long long *__restrict__ rowF,
                                                           unsigned long long *__restrict__ rowW,
                                                           unsigned int *__restrict__ cnt, int ncnt, SeedPack sp, int dtype,
                                                           const void *__restrict__ data, double t0, double t1,
                                                           unsigned int *__restrict__ seed_ok) {
    __shared__ unsigned int s_bad[64], s_has[64];
    const int64_t tyi = blockIdx.x % t.nty, tzi = blockIdx.x / t.nty;
    if (threadIdx.x < 64) {
        s_bad[threadIdx.x] = 0u;
        s_has[threadIdx.x] = 0u;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ncnt) cnt[threadIdx.x] = 0u; // the round counters (k_flood_block_apply appends)
    const int wx = (int)t.wx, per_z = TY * wx, nw = TZ * per_z;
    const int64_t tile0 = (int64_t)blockIdx.x * t.wx;
    if (FRESH) {
        if ((int)threadIdx.x < wx) {
            dirty[tile0 + threadIdx.x] = 0;
            dirty1[tile0 + threadIdx.x] = 0;
        }
        if (blockIdx.x == 0) {
            __shared__ unsigned int s_ok;
            if (threadIdx.x == 0) s_ok = 0u;
            __syncthreads();
            if ((int)threadIdx.x < sp.n) {
                const int64_t x = sp.xyz[threadIdx.x][0], y = sp.xyz[threadIdx.x][1], z = sp.xyz[threadIdx.x][2];
                const double v = load_as_double(data, dtype, (z * t.dy + y) * t.dx + x);
                if (v >= t0 && v <= t1) {
                    atomicOr(&cand[(z * t.dy + y) * t.wx + (x >> 6)], 1ull << (x & 63));
                    atomicOr(&s_ok, 1u << threadIdx.x);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) *seed_ok = s_ok;
        }
    }
    __syncthreads();
    const unsigned long long last = (t.dx & 63) ? (1ull << (t.dx & 63)) - 1ull : ~0ull;
    const unsigned long long one = bk.bxs == 64 ? ~0ull : ((1ull << bk.bxs) - 1ull);
    for (int i0 = threadIdx.x; i0 < nw; i0 += 4 * 256) { // four loads in flight per lane: the kernel is latency-bound
        unsigned long long cv[4], rv[4];
        int64_t wi[4];
        int tx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 256;
            const int tz = i / per_z, rem = i - tz * per_z, ty = rem / wx;
            tx[u] = rem - ty * wx;
            const int64_t z = tzi * TZ + tz, y = tyi * TY + ty;
            wi[u] = (i < nw && z < t.dz && y < t.dy) ? (z * t.dy + y) * t.wx + tx[u] : -1;
            cv[u] = wi[u] >= 0 ? cand[wi[u]] : 0ull;
            rv[u] = 0ull;
            if (!FRESH && wi[u] >= 0 && (dirty[tile0 + tx[u]] & 1u)) rv[u] = reached[wi[u]]; // only freshly seeded / marked tiles start a coarse flood
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (wi[u] < 0) continue;
            const unsigned long long miss = ~cv[u] & (tx[u] == wx - 1 ? last : ~0ull); // in-bounds voxels that are no candidates
            for (int q = 0; q < bk.q; q++) {
                if ((miss >> (bk.bxs * q)) & one) s_bad[tx[u] * bk.q + q] = 1u;
                else if (!FRESH && ((rv[u] >> (bk.bxs * q)) & one)) s_has[tx[u] * bk.q + q] = 1u;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int b = threadIdx.x, btx = b / bk.q;
        const bool in = btx < wx && (int64_t)btx * 64 + (int64_t)bk.bxs * (b % bk.q) < t.dx; // the block exists
        const unsigned long long f = __ballot(in && !s_bad[b]);
        const unsigned long long h = __ballot(in && s_has[b]);
        if (threadIdx.x == 0) {
            rowF[blockIdx.x] = f;
            rowW[blockIdx.x] = FRESH ? 0ull : (f & h);
        }
    }
}

// SEEDED: the accepted seeds of `sp` (mask *seed_ok) that sit in an all-candidate block start the coarse flood
template <int CONN, bool SEEDED>
__global__ __launch_bounds__(CT) void k_flood_coarse(Tiles t, Blocks bk, const unsigned long long *__restrict__ rowF,
                                                     unsigned long long *__restrict__ rowW, SeedPack sp,
                                                     const unsigned int *__restrict__ seed_ok) {
    __shared__ unsigned long long sR[CROWS_MAX];
    __shared__ unsigned int vote[2];
    const int nty = (int)t.nty, ntz = (int)t.ntz, hy = nty + 2;
    const int nrows = nty * ntz, nh = hy * (ntz + 2);
    unsigned long long F[CRP], R[CRP];
    int me[CRP];
#pragma unroll
    for (int k = 0; k < CRP; k++) { // issue the loads before the LDS clear
        const int j = threadIdx.x + k * (int)blockDim.x;
        F[k] = j < nrows ? rowF[j] : 0ull;
        R[k] = (j < nrows && !SEEDED) ? rowW[j] : 0ull;
        me[k] = j < nrows ? (j / nty + 1) * hy + (j % nty) + 1 : 0;
    }
    for (int i = threadIdx.x; i < nh; i += blockDim.x) sR[i] = 0ull;
    if (threadIdx.x == 0) vote[0] = 0u;
    __syncthreads();
    if (SEEDED) {
        if ((int)threadIdx.x < sp.n && (*seed_ok >> threadIdx.x & 1u)) {
            const int64_t x = sp.xyz[threadIdx.x][0], y = sp.xyz[threadIdx.x][1], z = sp.xyz[threadIdx.x][2];
            const int j = (int)((z / TZ) * nty + y / TY);
            const unsigned long long bit = 1ull << (x / bk.bxs);
            if (rowF[j] & bit) atomicOr(&sR[(j / nty + 1) * hy + (j % nty) + 1], bit);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CRP; k++) R[k] = sR[me[k]]; // (unused slots read the all-zero corner row)
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < CRP; k++)
        if (R[k]) {
            R[k] = fill_runs(R[k], F[k]);
            sR[me[k]] = R[k];
        }
    __syncthreads();
    for (int it = 0; it < 65536; it++) {
        bool changed = false;
#pragma unroll
        for (int k = 0; k < CRP; k++) {
            if (!(F[k] & ~R[k])) continue; // nothing left to gain in this row (also skips the unused slots)
            unsigned long long nb = 0;
#pragma unroll
            for (int kk = -1; kk <= 1; kk++)
#pragma unroll
                for (int jj = -1; jj <= 1; jj++) {
                    const int nz = (kk != 0) + (jj != 0);
                    const bool wide = CONN == 26 || (CONN == 18 && nz <= 1) || (CONN == 6 && nz == 0);
                    const bool mid = (CONN == 18 && nz == 2) || (CONN == 6 && nz == 1);
                    if (!wide && !mid) continue;
                    const unsigned long long n = sR[me[k] + kk * hy + jj];
                    nb |= wide ? (n | (n << 1) | (n >> 1)) : n;
                }
            const unsigned long long nr = fill_runs(R[k] | (nb & F[k]), F[k]);
            if (nr != R[k]) {
                R[k] = nr;
                sR[me[k]] = nr;
                changed = true;
            }
        }
        if (__any(changed) && (threadIdx.x & 63) == 0) vote[it & 1] = 1u;
        if (threadIdx.x == 0) vote[(it + 1) & 1] = 0u;
        __syncthreads();
        if (!vote[it & 1]) break;
    }
#pragma unroll
    for (int k = 0; k < CRP; k++) {
        const int j = threadIdx.x + k * (int)blockDim.x;
        if (j < nrows) rowW[j] = R[k];
    }
}

// reached = cand for the wholly reached blocks; the tiles around them that are not wholly reached themselves (and the
// seeded tiles that are not whole) go straight onto the round-0 list: a dirty byte is enlisted by whoever sets its
// ENLISTED bit first.
__device__ __forceinline__ void coarse_enlist(uint8_t *dirty, int64_t tile, unsigned int *list, unsigned int *count) {
    unsigned int *wp = (unsigned int *)(dirty + (tile & ~(int64_t)3));
    const unsigned int sh = 8 * (unsigned int)(tile & 3);
    const unsigned int old = atomicOr(wp, ENLISTED << sh);
    if (!(old & ((ENLISTED | CLOSED) << sh))) list[atomicAdd(count, 1u)] = (unsigned int)tile;
}

// FRESH: every word of the plane is written (whole blocks: their candidates; the accepted seeds' bits; zero elsewhere), so
// the plane needs no clearing pass before the flood
template <bool FRESH>
__global__ __launch_bounds__(256) void k_flood_block_apply(Tiles t, Blocks bk, const unsigned long long *__restrict__ cand,
                                                           unsigned long long *__restrict__ reached,
                                                           const unsigned long long *__restrict__ rowW, uint8_t *dirty,
                                                           unsigned int *__restrict__ list, unsigned int *count, SeedPack sp,
                                                           const unsigned int *__restrict__ seed_ok) {
    __shared__ unsigned int s_chg[64];
    __shared__ unsigned int s_seeds; // FRESH: the accepted seeds that sit in this tile row
    const int64_t tyi = blockIdx.x % t.nty, tzi = blockIdx.x / t.nty;
    const int wx = (int)t.wx, per_z = TY * wx, nw = TZ * per_z;
    const int64_t tile0 = (int64_t)blockIdx.x * t.wx;
    const unsigned long long whole = rowW[blockIdx.x];
    const unsigned int qm = (1u << bk.q) - 1u;
    if (threadIdx.x < 64) s_chg[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_seeds = 0u;
    __syncthreads();
    if (FRESH) {
        if ((int)threadIdx.x < sp.n && (*seed_ok >> threadIdx.x & 1u) && sp.xyz[threadIdx.x][2] / TZ == tzi &&
            sp.xyz[threadIdx.x][1] / TY == tyi) {
            s_chg[sp.xyz[threadIdx.x][0] >> 6] = 1u; // a seed starts its tile (and wakes the tiles around it)
            atomicOr(&s_seeds, 1u << threadIdx.x);
        }
    } else if ((int)threadIdx.x < wx && (dirty[tile0 + threadIdx.x] & 1u)) {
        const unsigned int ex = block_exist(t, bk, threadIdx.x);
        if (((unsigned int)(whole >> (bk.q * threadIdx.x)) & ex) == ex) {
            dirty[tile0 + threadIdx.x] = 0; // final: a wholly reached tile has nothing to gain
            // ... but it was marked because bits arrived in it from OUTSIDE the flood (a seed, a neighbour slab's plane
            // OR-ed into a halo slice): even when those bits already fill the tile, nobody has told its neighbours yet
            s_chg[threadIdx.x] = 1u;
        } else {
            coarse_enlist(dirty, tile0 + threadIdx.x, list, count);
        }
    }
    if (!FRESH && !whole) return; // uniform
    __syncthreads();
    const unsigned int ok = FRESH ? s_seeds : 0u;
    for (int i0 = threadIdx.x; i0 < nw; i0 += 4 * 256) { // four word pairs in flight per lane
        unsigned long long cv[4], rv[4], bm[4];
        int64_t wi[4];
        int tx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 256;
            const int tz = i / per_z, rem = i - tz * per_z, ty = rem / wx;
            tx[u] = rem - ty * wx;
            const int64_t z = tzi * TZ + tz, y = tyi * TY + ty;
            const unsigned int wq = (unsigned int)(whole >> (bk.q * tx[u])) & qm;
            wi[u] = (i < nw && (FRESH || wq) && z < t.dz && y < t.dy) ? (z * t.dy + y) * t.wx + tx[u] : -1;
            bm[u] = wq ? block_mask(bk, wq) : 0ull;
            cv[u] = (wi[u] >= 0 && wq) ? cand[wi[u]] : 0ull;
            rv[u] = (!FRESH && wi[u] >= 0) ? reached[wi[u]] : 0ull;
            if (FRESH && wi[u] >= 0 && ok) { // the accepted seeds of this word
                const int64_t z2 = tzi * TZ + tz, y2 = tyi * TY + ty;
                for (int n = 0; n < sp.n; n++)
                    if ((ok >> n & 1u) && sp.xyz[n][2] == z2 && sp.xyz[n][1] == y2 && (sp.xyz[n][0] >> 6) == tx[u])
                        rv[u] |= 1ull << (sp.xyz[n][0] & 63);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (wi[u] < 0) continue;
            const unsigned long long nr = rv[u] | (cv[u] & bm[u]);
            if (FRESH) {
                reached[wi[u]] = nr;
                if (bm[u]) s_chg[tx[u]] = 1u;
            } else if (nr != rv[u]) { // the tile gained voxels: its neighbours (and its own open part) must look again
                reached[wi[u]] = nr;
                s_chg[tx[u]] = 1u;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * wx; i += 256) {
        const int txi = i / 27, d = i - txi * 27;
        if (!s_chg[txi]) continue;
        const int64_t nz = tzi + d / 9 - 1, ny = tyi + (d / 3) % 3 - 1, nx = txi + d % 3 - 1;
        if (nz < 0 || nz >= t.ntz || ny < 0 || ny >= t.nty || nx < 0 || nx >= t.wx) continue;
        const unsigned int ex = block_exist(t, bk, nx);
        if (((unsigned int)(rowW[nz * t.nty + ny] >> (bk.q * nx)) & ex) == ex) continue; // wholly reached: nothing to gain
        coarse_enlist(dirty, (nz * t.nty + ny) * t.wx + nx, list, count);
    }
}

// ---- floodfill_auto_threshold: edge planes --------------------------------------------------------------------
// floodfill_py.rs:32-35: a voxel of value v lets the flood step to a 6-neighbour whose value lies in
// [ceil(v * (1 - p)), floor(v * (1 + p))], both f32 products cast to i16 the way Rust's `as` does (saturating, NaN -> 0),
// provided that neighbour's `out` byte is not `fill` yet.  One lane per 8 source voxels -> one byte in each of the six
// planes (+x, -x, +y, -y, +z, -z; bit = SOURCE voxel).
__device__ __forceinline__ int sat_i16(float f) {
    if (f != f) return 0;
    if (f <= -32768.0f) return -32768;
    if (f >= 32767.0f) return 32767;
    return (int)f;
}
__global__ __launch_bounds__(256) void k_flood_edges_auto(const int16_t *__restrict__ data, const uint8_t *__restrict__ out,
                                                          Tiles t, float p, uint8_t fill, uint8_t *__restrict__ edges) {
    const int64_t bpr = t.wx * 8, rows = t.dz * t.dy, total = rows * bpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float lo_f = 1.0f - p, hi_f = 1.0f + p;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t row = i / bpr, q = i - row * bpr;
        const int64_t z = row / t.dy, y = row - z * t.dy;
        unsigned m[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        const int64_t base = row * t.dx;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int64_t x = q * 8 + e;
            if (x >= t.dx) break;
            const float val = (float)data[base + x];
            const int t0 = sat_i16(ceilf(val * lo_f)), t1 = sat_i16(floorf(val * hi_f));
            const int64_t off[6] = {1, -1, t.dx, -t.dx, t.dy * t.dx, -t.dy * t.dx};
            const bool ok[6] = {x + 1 < t.dx, x > 0, y + 1 < t.dy, y > 0, z + 1 < t.dz, z > 0};
#pragma unroll
            for (int d = 0; d < 6; d++) {
                if (!ok[d]) continue;
                const int64_t n = base + x + off[d];
                const int nv = data[n];
                if (out[n] != fill && nv >= t0 && nv <= t1) m[d] |= 1u << e;
            }
        }
        const int64_t plane = total;
#pragma unroll
        for (int d = 0; d < 6; d++) edges[d * plane + i] = (uint8_t)m[d];
    }
}

// seeds whose bits are set no matter what the data says (floodfill.rs:21, floodfill_py.rs:30: `out[seed] = fill` and the
// seed is expanded unconditionally); `cand` may be NULL (directed floods have no candidate plane)
__global__ void k_flood_seed_forced(Tiles t, const int64_t *__restrict__ seeds, int64_t nseeds,
                                    unsigned long long *__restrict__ cand, unsigned long long *__restrict__ reached,
                                    uint8_t *__restrict__ dirty) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nseeds) return;
    const int64_t x = seeds[3 * n], y = seeds[3 * n + 1], z = seeds[3 * n + 2];
    const int64_t w = (z * t.dy + y) * t.wx + (x >> 6);
    const unsigned long long bit = 1ull << (x & 63);
    if (cand) atomicOr(&cand[w], bit);
    atomicOr(&reached[w], bit);
    mark_tile_nbhd(t, dirty, z / TZ, y / TY, x >> 6);
}

// ---- apply: out[v] = fill where reached ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_flood_apply(Tiles t, const uint8_t *__restrict__ reached, T *__restrict__ out,
                                                     T fill) {
    const int64_t bpr = t.wx * 8;
    const int64_t total = t.dz * t.dy * bpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool aligned8 = ((uintptr_t)out & 7) == 0 && (t.dx & 7) == 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned m = reached[i];
        if (!m) continue;
        const int64_t row = i / bpr, q = i - row * bpr;
        T *o = out + row * t.dx + q * 8;
        if (sizeof(T) == 1 && m == 0xffu && aligned8) { // inside a filled region: one 8-byte store
            *reinterpret_cast<unsigned long long *>(o) = 0x0101010101010101ull * (unsigned long long)(uint8_t)fill;
            continue;
        }
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (m >> e & 1u) o[e] = fill; // reached bits never exist beyond dx
    }
}

// byte targets with dx % 64 == 0 (rows are whole words, the bit volume is one flat array): one lane = 16 voxels =
// two bytes of reached bits -> one 16-byte store (a read-modify-write of the lane's own 16 bytes on a partial chunk)
__global__ __launch_bounds__(256) void k_flood_apply16(const uint16_t *__restrict__ reached, int64_t nchunks,
                                                       uint4 *__restrict__ out, uint8_t fill) {
    const int64_t stride = (int64_t)gridDim.x * 1024;
    const unsigned int f4 = 0x01010101u * fill;
    // four chunks per lane, 256 apart: the four bit loads (and the loads of the partial chunks' bytes) are in flight
    // together, and every store instruction of a wave writes 1 KB of consecutive bytes
    for (int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; i0 < nchunks; i0 += stride) {
        unsigned int m[4];
        uint4 o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t i = i0 + u * 256;
            m[u] = i < nchunks ? reached[i] : 0u;
            o[u] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (m[u] && m[u] != 0xffffu) o[u] = out[i0 + u * 256];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!m[u]) continue;
            uint4 v = make_uint4(f4, f4, f4, f4);
            if (m[u] != 0xffffu) {
                // bit e of m -> byte e: spread 4 bits to 4 byte masks
                auto blend = [&](unsigned int old, unsigned int bits4) {
                    const unsigned int sel = ((bits4 & 1u) * 0xffu) | ((bits4 >> 1 & 1u) * 0xff00u) |
                                             ((bits4 >> 2 & 1u) * 0xff0000u) | ((bits4 >> 3 & 1u) * 0xff000000u);
                    return (old & ~sel) | (f4 & sel);
                };
                v.x = blend(o[u].x, m[u] & 15u);
                v.y = blend(o[u].y, m[u] >> 4 & 15u);
                v.z = blend(o[u].z, m[u] >> 8 & 15u);
                v.w = blend(o[u].w, m[u] >> 12 & 15u);
            }
            out[i0 + u * 256] = v;
        }
    }
}

// out[v] = fill AND mask[v] = select where reached: floodfill_threshold's `out` plus the caller's
// `mask[out_mask.astype(bool)] = 254` (styles.py:3214) in one pass over the reached bits
__global__ __launch_bounds__(256) void k_flood_apply2(Tiles t, const uint8_t *__restrict__ reached, uint8_t *__restrict__ out,
                                                      uint8_t fill, uint8_t *__restrict__ mask, uint8_t select) {
    const int64_t bpr = t.wx * 8;
    const int64_t total = t.dz * t.dy * bpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool aligned8 = (((uintptr_t)out | (uintptr_t)mask) & 7) == 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned m = reached[i];
        if (!m) continue;
        const int64_t row = i / bpr, q = i - row * bpr;
        const int64_t base = row * t.dx + q * 8;
        if (m == 0xffu && ((t.dx & 7) == 0) && aligned8) { // inside a filled region: two 8-byte stores
            *reinterpret_cast<unsigned long long *>(out + base) = 0x0101010101010101ull * fill;
            *reinterpret_cast<unsigned long long *>(mask + base) = 0x0101010101010101ull * select;
            continue;
        }
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (m >> e & 1u) {
                out[base + e] = fill;
                mask[base + e] = select;
            }
    }
}

__global__ __launch_bounds__(256) void k_flood_count(const unsigned long long *__restrict__ bits, int64_t nwords,
                                                     unsigned long long *__restrict__ total) {
    unsigned long long s = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) s += __popcll(bits[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
}

static inline int grid_for(int64_t n) {
    const int64_t b = ivx::cdiv(n, 256);
    return (int)(b < 1 ? 1 : (b < 32768 ? b : 32768));
}

template <typename T>
static int run_candidates(const Tiles &t, const void *data, double t0, double t1, const uint8_t *bar, int bar_mode,
                          double fill, uint8_t *cand, hipStream_t st) {
    const int64_t total = t.dz * t.dy * t.wx * 8;
    if (!total) return IVX_OK;
    if (sizeof(T) == 2 && t.dx % 64 == 0 && (((uintptr_t)data | (uintptr_t)bar | (uintptr_t)cand) & 15) == 0) {
        // rows are whole words, so the bit volume is one flat array of 16-voxel chunks
        const int64_t nchunks = t.dz * t.dy * t.dx / 16;
        const int64_t blocks = ivx::cdiv(nchunks, 256);
        const int gg = (int)(blocks < 16384 ? blocks : 16384);
        const short8_t *d8 = (const short8_t *)data;
        const uint4 *b4 = (const uint4 *)bar;
        uint16_t *c16 = (uint16_t *)cand;
        if (bar_mode == 0) hipLaunchKernelGGL((k_flood_candidates16<T, 0>), dim3(gg), dim3(256), 0, st, d8, b4, nchunks, t0, t1, fill, c16);
        else if (bar_mode == 1) hipLaunchKernelGGL((k_flood_candidates16<T, 1>), dim3(gg), dim3(256), 0, st, d8, b4, nchunks, t0, t1, fill, c16);
        else hipLaunchKernelGGL((k_flood_candidates16<T, 2>), dim3(gg), dim3(256), 0, st, d8, b4, nchunks, t0, t1, fill, c16);
        IVX_LAUNCH_CHECK();
        return IVX_OK;
    }
    const int g = grid_for(total);
    if (bar_mode == 0)
        hipLaunchKernelGGL((k_flood_candidates<T, 0>), dim3(g), dim3(256), 0, st, (const T *)data, bar, t, t0, t1, fill, cand);
    else if (bar_mode == 1)
        hipLaunchKernelGGL((k_flood_candidates<T, 1>), dim3(g), dim3(256), 0, st, (const T *)data, bar, t, t0, t1, fill, cand);
    else
        hipLaunchKernelGGL((k_flood_candidates<T, 2>), dim3(g), dim3(256), 0, st, (const T *)data, bar, t, t0, t1, fill, cand);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

} // namespace

extern "C" int ivx_flood_bits_bytes(const ivx_flood_plan *p, size_t *nbytes) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    *nbytes = al256((size_t)(t.dz * t.dy * t.wx) * 8);
    return IVX_OK;
}

extern "C" int ivx_flood_scratch_bytes(const ivx_flood_plan *p, size_t *nbytes) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    *nbytes = make_fscratch(t).total;
    return IVX_OK;
}

// strct (sshape dims <= 3 each, centre offset dim/2: floodfill.rs:108-110) -> 27-bit mask in 3x3x3 offset space
extern "C" int ivx_flood_strct_bits(const uint8_t *strct, const int64_t sshape[3], uint32_t *bits) {
    IVX_REQUIRE(strct && sshape, IVX_EINVAL, "flood: strct is NULL");
    uint32_t b = 0;
    for (int a = 0; a < 3; a++)
        IVX_REQUIRE(sshape[a] >= 1 && sshape[a] <= 3, IVX_EINVAL, "flood: structuring element dims must be 1..3 (got %lld)",
                    (long long)sshape[a]);
    const int64_t oz = sshape[0] / 2, oy = sshape[1] / 2, ox = sshape[2] / 2;
    for (int64_t kk = 0; kk < sshape[0]; kk++)
        for (int64_t jj = 0; jj < sshape[1]; jj++)
            for (int64_t ii = 0; ii < sshape[2]; ii++)
                if (strct[(kk * sshape[1] + jj) * sshape[2] + ii]) {
                    const int64_t dz = kk - oz + 1, dy = jj - oy + 1, dx = ii - ox + 1;
                    b |= 1u << (dz * 9 + dy * 3 + dx);
                }
    *bits = b;
    return IVX_OK;
}

extern "C" int ivx_dev_flood_candidates(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                                        const uint8_t *barrier, int barrier_mode, double fill, uint64_t *cand,
                                        void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    IVX_REQUIRE(barrier_mode >= 0 && barrier_mode <= 2, IVX_EINVAL, "flood: barrier_mode");
    IVX_REQUIRE(barrier_mode != 1 || barrier, IVX_EINVAL, "flood: barrier array missing");
    hipStream_t st = ivx::S(stream);
    switch (dtype) {
    case IVX_I16: return run_candidates<int16_t>(t, data, t0, t1, barrier, barrier_mode, fill, (uint8_t *)cand, st);
    case IVX_U8: return run_candidates<uint8_t>(t, data, t0, t1, barrier, barrier_mode, fill, (uint8_t *)cand, st);
    case IVX_U16: return run_candidates<uint16_t>(t, data, t0, t1, barrier, barrier_mode, fill, (uint8_t *)cand, st);
    case IVX_F64: return run_candidates<double>(t, data, t0, t1, barrier, barrier_mode, fill, (uint8_t *)cand, st);
    }
    ivx::set_error("flood: unsupported dtype %d", dtype);
    return IVX_EINVAL;
}

extern "C" int ivx_dev_flood_seed(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                                  const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand, uint64_t *reached,
                                  void *scratch_, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    for (int64_t n = 0; n < nseeds; n++) {
        const int64_t x = seeds_xyz[3 * n], y = seeds_xyz[3 * n + 1], z = seeds_xyz[3 * n + 2];
        IVX_REQUIRE(x >= 0 && y >= 0 && z >= 0 && x < t.dx && y < t.dy && z < t.dz, IVX_ERANGE,
                    "flood: seed (%lld,%lld,%lld) outside volume (%lld,%lld,%lld) [x,y,z]", (long long)x, (long long)y,
                    (long long)z, (long long)t.dx, (long long)t.dy, (long long)t.dz);
    }
    const FScratch s = make_fscratch(t);
    char *scr = (char *)scratch_;
    hipStream_t st = ivx::S(stream);
    ivx::ccl_invalidate(scratch_); // a seed may add a bit to the candidate plane
    if (nseeds <= 16) { // the usual case (one click = one seed): seeds travel as kernel arguments, no staging, no sync
        SeedPack sp;
        sp.n = (int)nseeds;
        for (int64_t n = 0; n < nseeds; n++)
            for (int q = 0; q < 3; q++) sp.xyz[n][q] = seeds_xyz[3 * n + q];
        unsigned long long *c = (unsigned long long *)cand, *r = (unsigned long long *)reached;
        uint8_t *dirty = (uint8_t *)(scr + s.off_dirty0);
        if (nseeds == 0) return IVX_OK;
        switch (dtype) {
        case IVX_I16: hipLaunchKernelGGL(k_flood_seed_args<int16_t>, dim3(1), dim3(64), 0, st, (const int16_t *)data, t, t0, t1, sp, c, r, dirty); break;
        case IVX_U8: hipLaunchKernelGGL(k_flood_seed_args<uint8_t>, dim3(1), dim3(64), 0, st, (const uint8_t *)data, t, t0, t1, sp, c, r, dirty); break;
        case IVX_U16: hipLaunchKernelGGL(k_flood_seed_args<uint16_t>, dim3(1), dim3(64), 0, st, (const uint16_t *)data, t, t0, t1, sp, c, r, dirty); break;
        case IVX_F64: hipLaunchKernelGGL(k_flood_seed_args<double>, dim3(1), dim3(64), 0, st, (const double *)data, t, t0, t1, sp, c, r, dirty); break;
        default: ivx::set_error("flood: unsupported dtype %d", dtype); return IVX_EINVAL;
        }
        IVX_LAUNCH_CHECK();
        return IVX_OK;
    }
    int64_t *d_seeds = (int64_t *)(scr + s.off_seeds);
    for (int64_t b = 0; b < nseeds; b += (int64_t)SEED_CHUNK) {
        const int64_t m = nseeds - b < (int64_t)SEED_CHUNK ? nseeds - b : (int64_t)SEED_CHUNK;
        IVX_HIP(hipMemcpyAsync(d_seeds, seeds_xyz + 3 * b, (size_t)m * 24, hipMemcpyHostToDevice, st));
        const int g = (int)ivx::cdiv(m, 256);
        unsigned long long *c = (unsigned long long *)cand, *r = (unsigned long long *)reached;
        uint8_t *dirty = (uint8_t *)(scr + s.off_dirty0);
        switch (dtype) {
        case IVX_I16: hipLaunchKernelGGL(k_flood_seed<int16_t>, dim3(g), dim3(256), 0, st, (const int16_t *)data, t, t0, t1, d_seeds, m, c, r, dirty); break;
        case IVX_U8: hipLaunchKernelGGL(k_flood_seed<uint8_t>, dim3(g), dim3(256), 0, st, (const uint8_t *)data, t, t0, t1, d_seeds, m, c, r, dirty); break;
        case IVX_U16: hipLaunchKernelGGL(k_flood_seed<uint16_t>, dim3(g), dim3(256), 0, st, (const uint16_t *)data, t, t0, t1, d_seeds, m, c, r, dirty); break;
        case IVX_F64: hipLaunchKernelGGL(k_flood_seed<double>, dim3(g), dim3(256), 0, st, (const double *)data, t, t0, t1, d_seeds, m, c, r, dirty); break;
        default: ivx::set_error("flood: unsupported dtype %d", dtype); return IVX_EINVAL;
        }
        IVX_LAUNCH_CHECK();
        IVX_HIP(hipStreamSynchronize(st)); // d_seeds is reused by the next chunk; seeds_xyz is pageable host memory
    }
    return IVX_OK;
}

// one launch instead of three memsets: reached plane = 0, scratch head (dirty flags, counters, queue) = 0, ring = EMPTY
__global__ __launch_bounds__(256) void k_flood_clear(uint4 *__restrict__ reached, int64_t n16, uint4 *__restrict__ head,
                                                     int64_t h16, uint4 *__restrict__ ring, int64_t r16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u), e = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16 + h16 + r16; i += stride) {
        if (i < n16) reached[i] = z;
        else if (i < n16 + h16) head[i - n16] = z;
        else ring[i - n16 - h16] = e;
    }
}

extern "C" int ivx_dev_flood_clear(const ivx_flood_plan *p, uint64_t *reached, void *scratch, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    const FScratch s = make_fscratch(t);
    const size_t nb = (size_t)(t.dz * t.dy * t.wx) * 8, rb = (size_t)s.qcap * 4;
    if ((nb | s.off_seeds | rb | (uintptr_t)reached | (uintptr_t)scratch | s.off_ring) & 15) { // odd sizes: plain memsets
        IVX_HIP(hipMemsetAsync(reached, 0, nb, ivx::S(stream)));
        IVX_HIP(hipMemsetAsync(scratch, 0, s.off_seeds, ivx::S(stream)));
        IVX_HIP(hipMemsetAsync((char *)scratch + s.off_ring, 0xff, rb, ivx::S(stream)));
    } else {
        const int64_t total = (int64_t)((nb + s.off_seeds + rb) / 16);
        if (total) {
            const int64_t blocks = ivx::cdiv(total, 256);
            hipLaunchKernelGGL(k_flood_clear, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, ivx::S(stream),
                               (uint4 *)reached, (int64_t)(nb / 16), (uint4 *)scratch, (int64_t)(s.off_seeds / 16),
                               (uint4 *)((char *)scratch + s.off_ring), (int64_t)(rb / 16));
            IVX_LAUNCH_CHECK();
        }
    }
    ivx::ccl_invalidate(scratch);
    return IVX_OK;
}

// ---- gate: background work that should start once the flood's busy rounds are over --------------------------------------
// ivx_dev_flood_arm_gate(scratch, word, value, below): in the next flood on `scratch` the first round that starts with
// fewer than `below` tiles stores `value` to the device word `word` (and so does the end of the flood, if no round did).  ivx_dev_gate_wait(word, value, timeout, stream) parks `stream` behind a one-wave
// kernel that polls the word (bounded: it gives up after `timeout_us`).  Together: work queued on a second, low-priority
// stream runs under the latency-bound tail of the flood instead of competing with its throughput-bound head.
struct GateArm {
    unsigned int *word;
    unsigned int value, below;
};
static std::map<const void *, GateArm> g_gates;
static std::mutex g_gates_mu;

__global__ void k_gate_wait(const unsigned int *word, unsigned int value, unsigned long long timeout_ticks) {
    if (threadIdx.x) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value) {
        if (wall_clock64() - t0 > timeout_ticks) break; // never hang: start late rather than never
        __builtin_amdgcn_s_sleep(32);
    }
}
__global__ void k_gate_set(unsigned int *word, unsigned int value) {
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// directed: `cand` = the six edge planes of ivx_dev_flood_edges_auto (rounds engine only: the coarse pass, the
// union-find escape and the persistent frontier all rely on symmetric adjacency)
// `fresh` (may be NULL): the flood starts from these seeds on a plane whose old contents are dead (ivx_dev_flood_grow) --
// the coarse pass then does the clearing and the seeding as well
struct Fresh {
    int dtype;
    const void *data;
    double t0, t1;
    SeedPack sp;
};
static int flood_mode() {
    static const int mode_env = [] {
        const char *e = getenv("IVX_FLOOD_MODE");
        if (e && !strcmp(e, "ccl")) return 0;
        if (e && !strcmp(e, "persistent")) return 2;
        return 1;
    }();
    return mode_env;
}
static bool coarse_ok(const Tiles &t, bool directed) {
    static const bool coarse_on = [] {
        const char *e = getenv("IVX_FLOOD_COARSE");
        return !(e && e[0] == '0');
    }();
    // standard structures only, tile grid small enough for one workgroup's LDS
    return coarse_on && !directed && t.conn != 0 && t.wx <= 64 && (t.nty + 2) * (t.ntz + 2) <= CROWS_MAX &&
           t.nty * t.ntz <= (int64_t)CT * CRP;
}
// a resident launch (k_flood_resident) whose last word has not been read yet, keyed by the scratch it runs on
struct ResPending {
    volatile unsigned long long *line;
    uint32_t tag;
    bool ccl_at_cap;
};
static std::map<const void *, ResPending> g_res_pending;
static std::mutex g_res_mu;
static int flood_run_impl(const ivx_flood_plan *p, const uint64_t *cand, bool directed, uint64_t *reached, void *scratch_,
                          int *rounds, void *stream, const Fresh *fresh = nullptr, bool linear = false, int resident = -1);
// Wait for the resident launch on `scratch_` (if any) and finish what it left: *late = 1 when `reached` was completed after
// the launch had ended (round cap -> union-find engine; timed-out barrier -> launches per round), i.e. when work the caller
// queued behind the launch has seen an incomplete plane and must be queued again.
static int flood_wait_impl(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, void *scratch_, int *rounds,
                           int *late, void *stream) {
    if (late) *late = 0;
    ResPending rp;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        auto it = g_res_pending.find(scratch_);
        if (it == g_res_pending.end()) return IVX_OK; // nothing pending: the flood completed inside its call
        rp = it->second;
        g_res_pending.erase(it);
    }
    hipStream_t st = ivx::S(stream);
    unsigned long long v = 0;
    bool got = false;
    for (long spins = 0; spins < 40000000L; spins++) {
        v = __atomic_load_n((const unsigned long long *)rp.line, __ATOMIC_ACQUIRE);
        if ((uint32_t)(v >> 56) == rp.tag && ((v >> 48) & 0xffull) != 0) { got = true; break; }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!got) {
        IVX_HIP(hipStreamSynchronize(st));
        v = __atomic_load_n((const unsigned long long *)rp.line, __ATOMIC_ACQUIRE);
        IVX_REQUIRE((uint32_t)(v >> 56) == rp.tag && ((v >> 48) & 0xffull) != 0, IVX_EHIP, "flood: the resident launch never reported");
    }
    const unsigned int status = (unsigned int)((v >> 48) & 0xffull);
    int done_rounds = (int)(v & 0xffffffffull);
    static const bool trace = getenv("IVX_FLOOD_TRACE") != nullptr;
    if (trace) fprintf(stderr, "ivx flood: resident launch ended with status %u after %d rounds\n", status, done_rounds);
    if (status == 2u && rp.ccl_at_cap) { // long, thin region: the union-find engine completes the components reached so far
        Tiles t;
        int rc = make_tiles(p, &t);
        if (rc) return rc;
        const FScratch s = make_fscratch(t);
        char *scr = (char *)scratch_;
        IVX_HIP(hipMemsetAsync(scr + s.off_dirty0, 0, (size_t)t.ntiles, st));
        IVX_HIP(hipMemsetAsync(scr + s.off_dirty1, 0, (size_t)t.ntiles, st));
        if ((rc = ivx::ccl_run(p, cand, reached, scratch_, st))) return rc;
        done_rounds += 1;
        if (late) *late = 1;
    } else if (status != 1u) { // a barrier timed out (or a cap without an escape): finish with one launch per round
        fprintf(stderr, "ivx: resident flood launch ended with status %u after %d rounds; finishing with launches per round\n", status,
                done_rounds);
        Tiles t;
        int rc = make_tiles(p, &t);
        if (rc) return rc;
        const FScratch s = make_fscratch(t);
        char *scr = (char *)scratch_;
        IVX_HIP(hipMemsetAsync(scr + s.off_dirty0, 1, (size_t)t.ntiles, st));
        IVX_HIP(hipMemsetAsync(scr + s.off_dirty1, 0, (size_t)t.ntiles, st));
        int more = 0;
        if ((rc = flood_run_impl(p, cand, false, reached, scratch_, &more, stream, nullptr, false, 0))) return rc;
        done_rounds += more;
        if (late) *late = 1;
    }
    if (rounds) *rounds = done_rounds;
    return IVX_OK;
}
// resident: -1 = as IVX_FLOOD_RESIDENT says, returns when the flood is complete; 0 = launches per round; 1 = resident launch
// allowed AND the call may return right behind it (ivx_dev_flood_grow_async: the caller picks the result up with
// ivx_dev_flood_wait)
static int flood_run_impl(const ivx_flood_plan *p, const uint64_t *cand, bool directed, uint64_t *reached, void *scratch_,
                          int *rounds, void *stream, const Fresh *fresh, bool linear, int resident) {
    if (linear) directed = true; // (arc planes: no coarse pass, no union-find escape, no persistent frontier)
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    if (rounds) *rounds = 0;
    hipStream_t st = ivx::S(stream);
    GateArm arm = {nullptr, 0u, 0u};
    {
        std::lock_guard<std::mutex> lk(g_gates_mu);
        auto it = g_gates.find(scratch_);
        if (it != g_gates.end()) {
            arm = it->second;
            g_gates.erase(it);
        }
    }
    // whatever path the flood takes, the gate opens at the latest when it returns
    struct GateGuard {
        GateArm &a;
        hipStream_t st;
        ~GateGuard() {
            if (a.word) hipLaunchKernelGGL(k_gate_set, dim3(1), dim3(64), 0, st, a.word, a.value);
        }
    } gate_guard = {arm, st};
    if (t.ntiles == 0) return IVX_OK;
    const FScratch s = make_fscratch(t);
    char *scr = (char *)scratch_;
    uint8_t *dirty[2] = {(uint8_t *)(scr + s.off_dirty0), (uint8_t *)(scr + s.off_dirty1)};
    unsigned int *cnt = (unsigned int *)(scr + s.off_cnt);
    // IVX_FLOOD_MODE: "rounds" (default) = tile frontier, one launch per round, with an escape to the union-find path
    // when a flood needs more than CCL_ESCAPE_ROUNDS rounds (serpentine / maze-like regions); "ccl" = run-based
    // union-find from the start (k_ccl.hip; no frontier, flat cost); "persistent" = tile frontier, single launch +
    // device queue.  Measured at 512^3 on the bench blob (20 rounds): rounds 0.74 ms, ccl 0.86 ms, persistent 1.4 ms.
    const int mode = directed ? 1 : flood_mode();
    constexpr int CCL_ESCAPE_ROUNDS = 48;
    if (mode == 0 && ivx::ccl_supported(p->strct_bits)) {
        if (rounds) *rounds = 1;
        // the dirty-tile list is not used by this path; keep it empty so a later frontier run starts clean
        IVX_HIP(hipMemsetAsync(dirty[0], 0, (size_t)t.ntiles, st));
        return ivx::ccl_run(p, cand, reached, scratch_, st);
    }
    const bool use_rounds = mode != 2;
    static const unsigned int max_spins = [] {
        const char *e = getenv("IVX_FLOOD_MAX_SPINS");
        return e ? (unsigned int)strtoul(e, nullptr, 10) : (1u << 20);
    }();
    if (!use_rounds) {
        Queue *q = (Queue *)(scr + s.off_queue);
        unsigned int *queued = (unsigned int *)(scr + s.off_queued);
        unsigned int *ring = (unsigned int *)(scr + s.off_ring);
        IVX_HIP(hipMemsetAsync(&q->done, 0, 128, st)); // done / abort / visits
        // tickets of the previous run that were never served must not shift this run's slots
        IVX_HIP(hipMemsetAsync(&q->head, 0, 256, st)); // head, tail (ring is all-EMPTY between runs)
        hipLaunchKernelGGL(k_flood_enqueue, dim3((unsigned)ivx::cdiv(t.ntiles, 256)), dim3(256), 0, st, t, dirty[0], q,
                           queued, ring, s.qcap - 1);
        IVX_LAUNCH_CHECK();
        const int64_t grid = t.ntiles < 1024 ? t.ntiles : 1024; // <= 4 workgroups per CU; no residency requirement
        hipLaunchKernelGGL(k_flood_persistent, dim3((unsigned)grid), dim3(NT), 0, st, t,
                           (const unsigned long long *)cand, (unsigned long long *)reached, q, queued, ring,
                           s.qcap - 1, max_spins);
        IVX_LAUNCH_CHECK();
        Queue h;
        IVX_HIP(hipMemcpyAsync(&h, q, sizeof(Queue), hipMemcpyDeviceToHost, st));
        IVX_HIP(hipStreamSynchronize(st));
        if (!h.abort && h.pending == 0) {
            if (rounds) *rounds = (int)h.visits;
            return IVX_OK;
        }
        // The frontier kernel bounds every spin and bails out instead of hanging.  `reached` is monotone, so the
        // partial result is valid: reset the queue, mark every tile dirty and finish with one launch per round.
        fprintf(stderr, "ivx: persistent flood frontier aborted (head=%u tail=%u pending=%u abort=%u visits=%u); "
                        "finishing in rounds mode\n", h.head, h.tail, h.pending, h.abort, h.visits);
        IVX_HIP(hipMemsetAsync(scr + s.off_queue, 0, s.off_seeds - s.off_queue, st));
        IVX_HIP(hipMemsetAsync(scr + s.off_ring, 0xff, (size_t)s.qcap * 4, st));
        IVX_HIP(hipMemsetAsync(dirty[0], 1, (size_t)t.ntiles, st));
        IVX_HIP(hipMemsetAsync(dirty[1], 0, (size_t)t.ntiles, st));
    }
    int total_rounds = 0;
    unsigned int *list[2] = {(unsigned int *)(scr + s.off_list0), (unsigned int *)(scr + s.off_list1)};
    // Counters live in a ring of RING dwords indexed by the round number: round r reads cnt[r % RING] (entries of its
    // list), appends to cnt[(r+1) % RING] and clears cnt[(r+2) % RING] for the round after -- no host-side resets, so
    // rounds can be queued back to back.  Every round reports the length of its list to a pinned progress line as it
    // starts; the host keeps `ahead` rounds queued beyond the newest round it has seen start (a wait + re-launch round
    // trip would leave the GPU idle for 35-45 us) and stops at the first round that starts empty.  The rounds queued
    // ahead at that point find empty lists: a few ~2 us launches that later work simply queues behind.
    constexpr int RING = 2 * BATCH;
    static const int ahead = [] {
        const char *e = getenv("IVX_FLOOD_BATCH");
        const int v = e ? atoi(e) : 3;
        return v < 1 ? 1 : (v > BATCH ? BATCH : v);
    }();
    // coarse pass (see k_flood_coarse): blocks of 16 / 32 / 64 x 16 x 16 voxels, the finest whose row fits one word
    if (coarse_ok(t, directed)) {
        static const int bxs_env = [] {
            const char *e = getenv("IVX_FLOOD_BLOCK"); // 16 / 32 / 64: force a coarser block (A/B measurements)
            return e ? atoi(e) : 0;
        }();
        Blocks bk;
        bk.bxs = 16;
        while (bk.bxs < 64 && ((64 / bk.bxs) * t.wx > 64 || bk.bxs < bxs_env)) bk.bxs *= 2;
        bk.q = 64 / bk.bxs;
        unsigned long long *rowF = (unsigned long long *)(scr + s.off_full), *rowW = (unsigned long long *)(scr + s.off_whole);
        unsigned int *seed_ok = (unsigned int *)(scr + s.off_status) + 8;
        const unsigned groups = (unsigned)(t.nty * t.ntz);
        SeedPack none;
        none.n = 0;
        int ct = 64; // one row of tiles per lane up to 1024 lanes, then up to CRP rows per lane (128..1024 lanes measured equal)
        while (ct < CT && ct < t.nty * t.ntz) ct *= 2;
        if (fresh) {
            hipLaunchKernelGGL(k_flood_block_flags<true>, dim3(groups), dim3(256), 0, st, t, bk, (unsigned long long *)cand,
                               (const unsigned long long *)reached, dirty[0], dirty[1], rowF, rowW, cnt, RES_CTL_AT + 4, fresh->sp,
                               fresh->dtype, fresh->data, fresh->t0, fresh->t1, seed_ok);
            IVX_LAUNCH_CHECK();
            if (t.conn == 26) hipLaunchKernelGGL((k_flood_coarse<26, true>), dim3(1), dim3(ct), 0, st, t, bk, rowF, rowW, fresh->sp, seed_ok);
            else if (t.conn == 18) hipLaunchKernelGGL((k_flood_coarse<18, true>), dim3(1), dim3(ct), 0, st, t, bk, rowF, rowW, fresh->sp, seed_ok);
            else hipLaunchKernelGGL((k_flood_coarse<6, true>), dim3(1), dim3(ct), 0, st, t, bk, rowF, rowW, fresh->sp, seed_ok);
            IVX_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_flood_block_apply<true>, dim3(groups), dim3(256), 0, st, t, bk, (const unsigned long long *)cand,
                               (unsigned long long *)reached, rowW, dirty[0], list[0], cnt, fresh->sp, seed_ok);
            IVX_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(k_flood_block_flags<false>, dim3(groups), dim3(256), 0, st, t, bk, (unsigned long long *)cand,
                               (const unsigned long long *)reached, dirty[0], dirty[1], rowF, rowW, cnt, RES_CTL_AT + 4, none, 0,
                               (const void *)nullptr, 0.0, 0.0, seed_ok);
            IVX_LAUNCH_CHECK();
            if (t.conn == 26) hipLaunchKernelGGL((k_flood_coarse<26, false>), dim3(1), dim3(ct), 0, st, t, bk, rowF, rowW, none, seed_ok);
            else if (t.conn == 18) hipLaunchKernelGGL((k_flood_coarse<18, false>), dim3(1), dim3(ct), 0, st, t, bk, rowF, rowW, none, seed_ok);
            else hipLaunchKernelGGL((k_flood_coarse<6, false>), dim3(1), dim3(ct), 0, st, t, bk, rowF, rowW, none, seed_ok);
            IVX_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_flood_block_apply<false>, dim3(groups), dim3(256), 0, st, t, bk, (const unsigned long long *)cand,
                               (unsigned long long *)reached, rowW, dirty[0], list[0], cnt, none, seed_ok);
            IVX_LAUNCH_CHECK();
        }
    } else {
        IVX_REQUIRE(!fresh, IVX_EINVAL, "flood: the fused start needs the coarse pass");
        IVX_HIP(hipMemsetAsync(cnt, 0, (RES_CTL_AT + 4) * 4, st));
        hipLaunchKernelGGL(k_flood_build_list, dim3((unsigned)ivx::cdiv(t.ntiles, 256)), dim3(256), 0, st, t, dirty[0], list[0], cnt);
        IVX_LAUNCH_CHECK();
    }
    volatile unsigned long long *line = nullptr;
    uint32_t tag = 0;
    if ((rc = ivx::progress_line(st, &line, &tag))) return rc;
    // IVX_FLOOD_RESIDENT=1: the rounds in one resident launch (opt-in: measured slower, see k_flood_resident)
    static const bool resident_env = [] {
        const char *e = getenv("IVX_FLOOD_RESIDENT");
        return e && e[0] == '1';
    }();
    if (!directed && resident != 0 && resident_env) {
        static int res_per_cu = 0, ncu = 0;
        if (!ncu) {
            int dev = 0, occ = 0, n = 0;
            IVX_HIP(hipGetDevice(&dev));
            IVX_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            IVX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_flood_resident, NT, 0));
            const char *e = getenv("IVX_FLOOD_RES_PER_CU"); // workgroups per compute unit (A/B; never more than fit)
            int want = e ? atoi(e) : 4;
            if (want < 1) want = 1;
            res_per_cu = occ < 1 ? 0 : (want < occ ? want : occ);
            ncu = n > 0 ? n : 1;
        }
        if (res_per_cu > 0) {
            const int64_t cap = (int64_t)res_per_cu * ncu;
            const unsigned rgrid = (unsigned)(t.ntiles < cap ? t.ntiles : cap);
            const bool ccl_cap = ivx::ccl_supported(p->strct_bits);
            ResCtl *ctl = (ResCtl *)(cnt + RES_CTL_AT);
            hipLaunchKernelGGL(k_flood_resident, dim3(rgrid), dim3(NT), 0, st, t, (const unsigned long long *)cand,
                               (unsigned long long *)reached, list[0], list[1], cnt, dirty[0], dirty[1], ctl,
                               (unsigned long long *)line, (unsigned int)tag, (unsigned int)RING,
                               ccl_cap ? (unsigned int)CCL_ESCAPE_ROUNDS : 0x00ffffffu, max_spins, arm.word, arm.value, arm.below);
            IVX_LAUNCH_CHECK();
            arm.word = nullptr; // the launch opens the gate itself
            {
                std::lock_guard<std::mutex> lk(g_res_mu);
                g_res_pending[scratch_] = ResPending{line, tag, ccl_cap};
            }
            if (resident == 1) return IVX_OK; // the caller collects the result (ivx_dev_flood_wait)
            return flood_wait_impl(p, cand, reached, scratch_, rounds, nullptr, stream);
        }
    }
    const unsigned grid_max = (unsigned)(t.ntiles < 1536 ? t.ntiles : 1536); // 6 workgroups per CU are resident (80 VGPRs)
    // The rounds queued ahead are sized by the newest list length the host has seen: the lists shrink towards the end, and
    // dispatching 1536 workgroups that find nothing costs ~3 us more per round than dispatching 128 (a list that turns out
    // longer than the grid is still served: workgroups stride over it).
    unsigned grid = grid_max;
    int64_t queued = 0; // rounds launched so far
    auto queue_round = [&]() -> int {
        const int r = (int)(queued % RING), cur = (int)(queued & 1);
        const unsigned int tr = (unsigned int)(tag << 24) | (unsigned int)((queued + 1) & 0xffffff);
        unsigned int *gw = arm.word; // every round carries the gate: the first one with a short list opens it
        if (linear)
            hipLaunchKernelGGL(k_flood_round_list<2>, dim3(grid), dim3(NT), 0, st, t, (const unsigned long long *)cand,
                               (unsigned long long *)reached, list[cur], cnt + r, dirty[cur], dirty[cur ^ 1], list[cur ^ 1],
                               cnt + (r + 1) % RING, cnt + (r + 2) % RING, (unsigned long long *)line, tr, gw, arm.value, arm.below);
        else if (directed)
            hipLaunchKernelGGL(k_flood_round_list<1>, dim3(grid), dim3(NT), 0, st, t, (const unsigned long long *)cand,
                               (unsigned long long *)reached, list[cur], cnt + r, dirty[cur], dirty[cur ^ 1], list[cur ^ 1],
                               cnt + (r + 1) % RING, cnt + (r + 2) % RING, (unsigned long long *)line, tr, gw, arm.value, arm.below);
        else
            hipLaunchKernelGGL(k_flood_round_list<0>, dim3(grid), dim3(NT), 0, st, t, (const unsigned long long *)cand,
                               (unsigned long long *)reached, list[cur], cnt + r, dirty[cur], dirty[cur ^ 1], list[cur ^ 1],
                               cnt + (r + 1) % RING, cnt + (r + 2) % RING, (unsigned long long *)line, tr, gw, arm.value, arm.below);
        IVX_LAUNCH_CHECK();
        queued++;
        return IVX_OK;
    };
    static const bool trace = getenv("IVX_FLOOD_TRACE") != nullptr;
    int64_t seen = 0; // rounds seen starting
    for (;;) {
        while (queued < seen + 1 + ahead)
            if ((rc = queue_round())) return rc;
        // wait for a round beyond `seen` to report
        unsigned long long v = 0;
        bool got = false;
        for (long spins = 0; spins < 20000000L; spins++) {
            v = __atomic_load_n((const unsigned long long *)line, __ATOMIC_ACQUIRE);
            if ((uint32_t)(v >> 56) == tag && (int64_t)((v >> 32) & 0xffffffull) > seen) { got = true; break; }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        if (!got) { // a few hundred ms without news: take the safe path once
            IVX_HIP(hipStreamSynchronize(st));
            v = __atomic_load_n((const unsigned long long *)line, __ATOMIC_ACQUIRE);
            IVX_REQUIRE((uint32_t)(v >> 56) == tag && (int64_t)((v >> 32) & 0xffffffull) > seen, IVX_EHIP,
                        "flood: the queued rounds never reported");
        }
        seen = (int64_t)((v >> 32) & 0xffffffull);
        const unsigned int n_list = (unsigned int)(v & 0xffffffffull);
        if (trace) fprintf(stderr, "ivx flood: round %lld starts with %u of %lld tiles (%lld queued)\n", (long long)seen, n_list,
                           (long long)t.ntiles, (long long)queued);
        if (n_list < arm.below) arm.word = nullptr; // that round has opened the gate: nothing left for the guard to do
        grid = n_list >= 384u ? grid_max : (n_list >= 96u ? (grid_max < 512u ? grid_max : 512u) : (grid_max < 128u ? grid_max : 128u));
        if (n_list == 0) { // converged: word 1 holds the last round that had work (if any, and if it is ours)
            const unsigned long long u = __atomic_load_n((const unsigned long long *)line + 1, __ATOMIC_ACQUIRE);
            total_rounds = (uint32_t)(u >> 56) == tag ? (int)((u >> 32) & 0xffffffull) : 0;
            break;
        }
        total_rounds = (int)seen;
        if (!directed && total_rounds >= CCL_ESCAPE_ROUNDS && ivx::ccl_supported(p->strct_bits)) {
            // long, thin region: stop paying one launch per tile hop -- every reached bit so far is correct, the
            // union-find path completes the components they belong to in one flat pass (the rounds already queued
            // keep flooding until then, which is harmless)
            IVX_HIP(hipMemsetAsync(dirty[0], 0, (size_t)t.ntiles, st));
            IVX_HIP(hipMemsetAsync(dirty[1], 0, (size_t)t.ntiles, st));
            if ((rc = ivx::ccl_run(p, cand, reached, scratch_, st))) return rc;
            total_rounds += 1;
            break;
        }
        IVX_REQUIRE(total_rounds < (1 << 24) - 64, IVX_EHIP, "flood: did not converge");
    }
    if (rounds) *rounds = total_rounds;
    return IVX_OK;
}

// ---- the IFT watershed's cost map, level by level, on bit planes (ivx_dev_ws_cost_levels) ------------------------------
// C(p) = min over paths from a marker of the largest arc |I(a) - I(b)| on the path.  {C <= c} is the set the markers reach
// through arcs of weight <= c: a flood on bit planes with arc planes instead of a candidate plane, and {C == c} is what
// level c adds to level c - 1.  The chaotic relaxation of the cost map (k_ws_relax) spends its time on the levels where the
// bulk of a noise volume connects (percolation: long winding paths, every tile revisited ~15 times with 16-bit costs in
// LDS); here those levels cost bit-parallel tile visits (64 voxels per lane and operation).  The caller stops after the
// bulk is in and hands the rest -- isolated pockets whose cost is decided by their own few arcs -- to the relaxation,
// which starts from exact costs and has nothing left to correct.
namespace {
template <typename MT>
__global__ __launch_bounds__(256) void k_wsa_seed(const MT *__restrict__ mk, int64_t n, unsigned long long *__restrict__ R) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long b = __ballot(p < n && mk[p] != 0);
    if ((threadIdx.x & 63) == 0 && p < n) R[p >> 6] = b;
}

// Arc weights once, as bytes: wx / wy / wz[p] = min(|I(p) - I(p + 1 / W / HW)|, 127), 127 also when the neighbour's linear
// index is >= n (levels stop far below 127: the caller caps them at 120).  Lane = 8 voxels; the ALU-heavy part of the arc
// planes (field extraction, absolute differences) then happens once instead of once per level.
__global__ __launch_bounds__(256) void k_wsa_weights(const uint16_t *__restrict__ I, int64_t n, int64_t W, int64_t HW,
                                                     unsigned long long *__restrict__ wts) {
    const int64_t nch = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nch; i += stride) {
        const int64_t p0 = i << 3;
        const bool hy = p0 + W < n, hz = p0 + HW < n;
        const uint4 v = *reinterpret_cast<const uint4 *>(I + p0);
        const uint4 vy = *reinterpret_cast<const uint4 *>(I + (hy ? p0 + W : p0));
        const uint4 vz = *reinterpret_cast<const uint4 *>(I + (hz ? p0 + HW : p0));
        const int next = p0 + 8 < n ? (int)I[p0 + 8] : -1000000;
        const unsigned int vw[4] = {v.x, v.y, v.z, v.w}, yw[4] = {vy.x, vy.y, vy.z, vy.w}, zw[4] = {vz.x, vz.y, vz.z, vz.w};
        unsigned long long bx = 0, by = 0, bz = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int iv = (int)((vw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
            const int nx = e < 7 ? (int)((vw[(e + 1) >> 1] >> (16 * ((e + 1) & 1))) & 0xffffu) : next;
            const int ny = (int)((yw[e >> 1] >> (16 * (e & 1))) & 0xffffu), nz = (int)((zw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
            const int dx = min(abs(iv - nx), 127), dy = hy ? min(abs(iv - ny), 127) : 127, dz = hz ? min(abs(iv - nz), 127) : 127;
            bx |= (unsigned long long)dx << (8 * e);
            by |= (unsigned long long)dy << (8 * e);
            bz |= (unsigned long long)dz << (8 * e);
        }
        wts[i] = bx;
        wts[nch + i] = by;
        wts[2 * nch + i] = bz;
    }
}

// the arc planes of level c from the weight bytes: lane = word = 8 x 8 bytes per direction; "byte <= c" for eight bytes at
// once: (x | 0x80) - (c + 1) keeps bit 7 exactly when x >= c + 1 (x, c < 128: no borrow between bytes), and the eight
// sign bits are gathered with one multiply
__device__ __forceinline__ unsigned long long le8(unsigned long long x, unsigned long long c1) {
    const unsigned long long t = (x | 0x8080808080808080ull) - c1;
    return ((~t & 0x8080808080808080ull) * 0x0002040810204081ull) >> 56;
}
// The arc planes of L consecutive levels c .. c + L - 1 in one pass over the weight bytes (level l's three planes at
// E + l * 3 * nwords): the weights are 3 bytes per voxel, a level's planes 3 bits -- one pass per level read 400 MB to write 48
// (84 us x 15 levels at 512^3, 0.67 ms x 12 at 1024^3).
template <int L>
__global__ __launch_bounds__(256) void k_wsa_planes(const unsigned long long *__restrict__ wts, int64_t nwords, int c,
                                                    unsigned long long *__restrict__ E) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    const int64_t nch = nwords * 8;
    unsigned long long c1[L];
#pragma unroll
    for (int l = 0; l < L; l++) c1[l] = (unsigned long long)(c + l + 1) * 0x0101010101010101ull;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(wts + d * nch + w * 8);
        unsigned long long m[L];
#pragma unroll
        for (int l = 0; l < L; l++) m[l] = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ulonglong2 x = src[q];
#pragma unroll
            for (int l = 0; l < L; l++) {
                m[l] |= le8(x.x, c1[l]) << (16 * q);
                m[l] |= le8(x.y, c1[l]) << (16 * q + 8);
            }
        }
#pragma unroll
        for (int l = 0; l < L; l++) E[((int64_t)l * 3 + d) * nwords + w] = m[l];
    }
}

// lane = word: would ONE relaxation step add a bit to this word?  Then its tile starts the level's flood.
__global__ __launch_bounds__(256) void k_wsa_frontier(Tiles t, const unsigned long long *__restrict__ R,
                                                      const unsigned long long *__restrict__ E, uint8_t *__restrict__ dirty) {
    const int64_t nwords = t.dz * t.dy * t.wx, hwx = t.dy * t.wx;
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    const unsigned long long *Ex = E, *Ey = E + nwords, *Ez = E + 2 * nwords;
    const unsigned long long r = R[w];
    if (r == ~0ull) return;
    const unsigned long long exp = Ex[w];
    unsigned long long nr = r;
    if (w > 0) nr |= (R[w - 1] & Ex[w - 1]) >> 63;
    if (w + 1 < nwords) nr |= ((R[w + 1] & 1ull) & (exp >> 63)) << 63;
    if (w - t.wx >= 0) nr |= R[w - t.wx] & Ey[w - t.wx];
    if (w + t.wx < nwords) nr |= R[w + t.wx] & Ey[w];
    if (w - hwx >= 0) nr |= R[w - hwx] & Ez[w - hwx];
    if (w + hwx < nwords) nr |= R[w + hwx] & Ez[w];
    nr |= ((nr & exp) << 1) | ((nr & (exp << 1)) >> 1); // one step along x inside the word is enough to see a gain
    if (nr != r) {
        const int64_t row = w / t.wx, txi = w - row * t.wx, z = row / t.dy, y = row - z * t.dy;
        dirty[((z / TZ) * t.nty + (y >> TY_LOG)) * t.wx + txi] = 1;
    }
}

// reached voxels so far (grid-stride, one atomic per workgroup)
__global__ __launch_bounds__(256) void k_wsa_count(const unsigned long long *__restrict__ R, int64_t nwords,
                                                   unsigned long long *__restrict__ count) {
    __shared__ unsigned long long s_part[4];
    unsigned long long mine = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) mine += (unsigned long long)__popcll(R[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (t) atomicAdd(count, t);
    }
}

// After the last level: snap[l] = the reached plane as level l left it (planes nested: a bit set at level l is set at every
// later one).  C[p] = the first level that has p; voxels no level reached keep their cost.  Lane = 16 voxels.
__global__ __launch_bounds__(256) void k_wsa_costs(const uint16_t *__restrict__ snap, int64_t nchunks, int64_t plane_chunks, int levels,
                                                   uint16_t *__restrict__ C) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        const unsigned int last = snap[(int64_t)(levels - 1) * plane_chunks + i];
        if (!last) continue;
        uint16_t *dst = C + i * 16;
        unsigned int lev[16];
#pragma unroll
        for (int e = 0; e < 16; e++) lev[e] = 0xffffu;
        unsigned int have = 0;
        for (int l = 0; l < levels && have != last; l++) {
            const unsigned int m = snap[(int64_t)l * plane_chunks + i];
            unsigned int nw = m & ~have;
            have |= m;
#pragma unroll
            for (int e = 0; e < 16; e++)
                if (nw >> e & 1u) lev[e] = (unsigned int)l;
        }
        if (last == 0xffffu) {
            reinterpret_cast<uint4 *>(dst)[0] = make_uint4(lev[0] | lev[1] << 16, lev[2] | lev[3] << 16, lev[4] | lev[5] << 16, lev[6] | lev[7] << 16);
            reinterpret_cast<uint4 *>(dst)[1] = make_uint4(lev[8] | lev[9] << 16, lev[10] | lev[11] << 16, lev[12] | lev[13] << 16, lev[14] | lev[15] << 16);
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++)
                if (last >> e & 1u) dst[e] = (uint16_t)lev[e];
        }
    }
}
} // namespace

// Levels 0, 1, 2, ... of the IFT cost map until `stop_frac` of the voxels are in (or `max_levels` are done): C[p] = level for
// every voxel reached (the others keep what the caller put there: 0xFFFF), *levels_done = number of levels completed,
// *reached_out = voxels with a final cost.  6-neighbour structure, scipy's linear-index neighbourhood; needs dx % 64 == 0
// and dy % 16 == 0 (IVX_EINVAL otherwise: the caller then runs its relaxation from the markers alone).
extern "C" int ivx_dev_ws_cost_levels(const uint16_t *I, int mdtype, const void *markers, int64_t dz, int64_t dy, int64_t dx,
                                      uint16_t *C, int max_levels, double stop_frac, int *levels_done, int64_t *reached_out,
                                      int64_t *rounds_out, void *stream) {
    IVX_REQUIRE(I && markers && C && dz > 0 && dy > 0 && dx > 0, IVX_EINVAL, "ws_cost_levels: bad arguments");
    IVX_REQUIRE(dx % 64 == 0 && dy % TY == 0, IVX_EINVAL, "ws_cost_levels: needs dx %% 64 == 0 and dy %% 16 == 0");
    IVX_REQUIRE(mdtype == IVX_I16 || mdtype == IVX_I8, IVX_EINVAL, "ws_cost_levels: markers must be int16 or int8");
    ivx_flood_plan plan;
    plan.dz = dz; plan.dy = dy; plan.dx = dx; plan.wx = dx / 64;
    plan.strct_bits = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 14) | (1u << 16) | (1u << 22);
    Tiles t;
    int rc = make_tiles(&plan, &t);
    if (rc) return rc;
    hipStream_t st = ivx::S(stream);
    const int64_t n = dz * dy * dx, nwords = n >> 6;
    const FScratch fs = make_fscratch(t);
    IVX_REQUIRE(max_levels >= 1, IVX_EINVAL, "ws_cost_levels: max_levels");
    if (max_levels > 120) max_levels = 120; // (the weight bytes saturate at 127)
    // workspace: R | arc planes Ex Ey Ez | weight bytes x y z | flood scratch | one snapshot of R per level
    constexpr int PL = 4; // levels whose arc planes are made by one pass over the weights
    const size_t pw = (size_t)nwords * 8, o_E = al256(pw), o_W = al256(o_E + (size_t)PL * 3 * pw), o_S = al256(o_W + 3 * (size_t)n);
    const size_t o_P = al256(o_S + fs.total);
    void *mem;
    if ((rc = ivx::ws_get_s(ivx::WS_WSA, st, o_P + (size_t)max_levels * pw + 256, &mem))) return rc;
    unsigned long long *R = (unsigned long long *)mem;
    unsigned long long *E_all = (unsigned long long *)((char *)mem + o_E); // PL levels x (Ex | Ey | Ez), nwords each
    unsigned long long *wts = (unsigned long long *)((char *)mem + o_W); // n bytes per direction
    char *scr = (char *)mem + o_S;
    char *snaps = (char *)mem + o_P;
    unsigned long long *d_count = (unsigned long long *)(scr + fs.off_status) + 2;
    IVX_HIP(hipMemsetAsync(scr, 0, fs.off_seeds, st)); // dirty flags, counters
    const unsigned gv = (unsigned)ivx::cdiv(n, 256), gw = (unsigned)ivx::cdiv(nwords, 256);
    if (mdtype == IVX_I16) hipLaunchKernelGGL(k_wsa_seed<int16_t>, dim3(gv), dim3(256), 0, st, (const int16_t *)markers, n, R);
    else hipLaunchKernelGGL(k_wsa_seed<int8_t>, dim3(gv), dim3(256), 0, st, (const int8_t *)markers, n, R);
    IVX_LAUNCH_CHECK();
    {
        const int64_t blocks = ivx::cdiv(n >> 3, 256);
        hipLaunchKernelGGL(k_wsa_weights, dim3((unsigned)(blocks < 32768 ? blocks : 32768)), dim3(256), 0, st, I, n, dx, dy * dx, wts);
        IVX_LAUNCH_CHECK();
    }
    int64_t rounds_total = 0, reached = 0;
    int c = 0;
    for (; c < max_levels; c++) {
        if (c % PL == 0) { // (weights above 127 saturate: levels beyond max_levels <= 120 are never asked for, their planes cost nothing extra)
            hipLaunchKernelGGL(k_wsa_planes<PL>, dim3(gw), dim3(256), 0, st, wts, nwords, c, E_all);
            IVX_LAUNCH_CHECK();
        }
        unsigned long long *E = E_all + (size_t)(c % PL) * 3 * nwords;
        hipLaunchKernelGGL(k_wsa_frontier, dim3(gw), dim3(256), 0, st, t, R, E, (uint8_t *)(scr + fs.off_dirty0));
        IVX_LAUNCH_CHECK();
        int rounds = 0;
        if ((rc = flood_run_impl(&plan, (const uint64_t *)E, true, (uint64_t *)R, scr, &rounds, stream, nullptr, true))) return rc;
        rounds_total += rounds;
        IVX_HIP(hipMemcpyAsync(snaps + (size_t)c * pw, R, pw, hipMemcpyDeviceToDevice, st));
        IVX_HIP(hipMemsetAsync(d_count, 0, 8, st));
        hipLaunchKernelGGL(k_wsa_count, dim3(gw < 1024 ? gw : 1024), dim3(256), 0, st, R, nwords, d_count);
        IVX_LAUNCH_CHECK();
        uint32_t seq, got[2] = {0, 0};
        if ((rc = ivx::mailbox_publish(d_count, 2, st, &seq))) return rc;
        if ((rc = ivx::mailbox_wait(seq, st, got, 2))) return rc;
        reached = (int64_t)(((uint64_t)got[1] << 32) | got[0]);
        if ((double)reached >= stop_frac * (double)n) {
            c++;
            break;
        }
    }
    {
        const int64_t nchunks = n / 16, blocks = ivx::cdiv(nchunks, 256);
        hipLaunchKernelGGL(k_wsa_costs, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, (const uint16_t *)snaps,
                           nchunks, (int64_t)(pw / 2), c, C);
        IVX_LAUNCH_CHECK();
    }
    if (levels_done) *levels_done = c;
    if (reached_out) *reached_out = reached;
    if (rounds_out) *rounds_out = rounds_total;
    return IVX_OK;
}

// ---- the scikit-image branch's cost map, level by level (ivx_dev_sk_cost_levels) ---------------------------------------
// There a path costs the largest image VALUE on it (markers cost their own value), so {C <= c} is what the markers of value
// <= c reach inside the candidate plane {I <= c}: the ordinary region-growing engine, coarse pass included.  The gradient
// of a windowed image is zero over everything the window saturates: level 0 alone is ~95 % of such a volume, one flood.
namespace {
// Eight lanes' bytes of bits -> one word in the first of them (lane & 7 == 0): three exchanges.
__device__ __forceinline__ unsigned long long gather_word8(uint32_t bits8, int lane) {
    unsigned long long w = (unsigned long long)bits8 << (8 * (lane & 7));
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)w, o, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(w >> 32), o, 64);
        w |= ((unsigned long long)hi << 32) | lo;
    }
    return w;
}
// The markers as a bit plane, once per call (lane = 8 voxels: one 16-byte load of int16 markers): a level's seeds are then
// marker & candidate & ~reached per WORD.  (Rounds 1 - 5 read the marker volume voxel by voxel at every level: 170 us x 3 at
// 512^3, 1.35 ms x 3 at 1024^3.)
template <typename MT>
__global__ __launch_bounds__(256) void k_ska_marker_bits(const MT *__restrict__ mk, int64_t n8, unsigned long long *__restrict__ mb) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; // (n8 is a multiple of 8: whole words, whole groups of eight lanes)
    uint32_t bits = 0;
    if (i < n8) {
        MT v[8];
        if (sizeof(MT) == 2) *reinterpret_cast<uint4 *>(v) = reinterpret_cast<const uint4 *>(mk)[i];
        else *reinterpret_cast<uint2 *>(v) = reinterpret_cast<const uint2 *>(mk)[i];
#pragma unroll
        for (int e = 0; e < 8; e++) bits |= (v[e] != 0 ? 1u : 0u) << e;
    }
    const unsigned long long w = gather_word8(bits, threadIdx.x & 63);
    if ((threadIdx.x & 7) == 0 && i < n8) mb[i >> 3] = w;
}
// ... and the candidate planes {I <= l} of the first L levels in one pass over the image (L <= 4)
template <int L>
__global__ __launch_bounds__(256) void k_ska_cands(const uint16_t *__restrict__ I, int64_t n8, int64_t nwords, unsigned long long *__restrict__ cand) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t bits[L];
#pragma unroll
    for (int l = 0; l < L; l++) bits[l] = 0;
    if (i < n8) {
        uint16_t v[8];
        *reinterpret_cast<uint4 *>(v) = reinterpret_cast<const uint4 *>(I)[i];
#pragma unroll
        for (int e = 0; e < 8; e++)
#pragma unroll
            for (int l = 0; l < L; l++) bits[l] |= (v[e] <= (uint16_t)l ? 1u : 0u) << e;
    }
#pragma unroll
    for (int l = 0; l < L; l++) {
        const unsigned long long w = gather_word8(bits[l], threadIdx.x & 63);
        if ((threadIdx.x & 7) == 0 && i < n8) cand[(int64_t)l * nwords + (i >> 3)] = w;
    }
}
// seeds of level c, lane = word
__global__ __launch_bounds__(256) void k_ska_seed_bits(Tiles t, const unsigned long long *__restrict__ mb, const unsigned long long *__restrict__ cand,
                                                       unsigned long long *__restrict__ R, uint8_t *__restrict__ dirty) {
    const int64_t nwords = t.dz * t.dy * t.wx;
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    const unsigned long long add = mb[w] & cand[w] & ~R[w];
    if (!add) return;
    R[w] |= add;
    const int64_t row = w / t.wx, txi = w - row * t.wx, z = row / t.dy, y = row - z * t.dy;
    mark_tile_nbhd(t, dirty, z / TZ, y / TY, txi);
}

// lane = word: a candidate bit that is not reached and has a reached neighbour under the structure -> its tile (and the
// tiles around it) start the level's flood
__global__ __launch_bounds__(256) void k_ska_frontier(Tiles t, const unsigned long long *__restrict__ cand,
                                                      const unsigned long long *__restrict__ R, uint8_t *__restrict__ dirty) {
    const int64_t nwords = t.dz * t.dy * t.wx;
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    const unsigned long long open = cand[w] & ~R[w];
    if (!open) return;
    const int64_t row = w / t.wx, txi = w - row * t.wx, z = row / t.dy, y = row - z * t.dy;
    unsigned long long nb = 0;
    for (int kk = 0; kk < 3; kk++)
        for (int jj = 0; jj < 3; jj++) {
            const uint32_t m3 = (t.strct >> (kk * 9 + jj * 3)) & 7u;
            if (!m3) continue;
            // voxel q reached => q + (kk-1, jj-1, ii-1) reached: the source row of this word is (z - (kk-1), y - (jj-1))
            const int64_t zs = z - (kk - 1), ys = y - (jj - 1);
            if (zs < 0 || zs >= t.dz || ys < 0 || ys >= t.dy) continue;
            const unsigned long long *rr = R + (zs * t.dy + ys) * t.wx;
            const unsigned long long c0 = rr[txi];
            const unsigned long long cl = txi > 0 ? rr[txi - 1] >> 63 : 0ull, cr = txi + 1 < t.wx ? rr[txi + 1] & 1ull : 0ull;
            if (m3 & 2u) nb |= c0;
            if (m3 & 4u) nb |= (c0 << 1) | cl;        // ii = 2: source bit x - 1
            if (m3 & 1u) nb |= (c0 >> 1) | (cr << 63); // ii = 0: source bit x + 1
        }
    if (nb & open) mark_tile_nbhd(t, dirty, z / TZ, y / TY, txi);
}
} // namespace

// Levels 0, 1, 2, ... of scikit-image's cost map (value-on-path minimax, lattice neighbours, any symmetric 3x3x3 structure)
// until `stop_frac` of the voxels are in or `max_levels` are done; C[p] = level for every voxel reached.  Needs dx % 64 == 0,
// 16-byte aligned image and markers.
extern "C" int ivx_dev_sk_cost_levels(const uint16_t *I, int mdtype, const void *markers, int64_t dz, int64_t dy, int64_t dx,
                                      const uint8_t strct[27], uint16_t *C, int max_levels, double stop_frac, int *levels_done,
                                      int64_t *reached_out, int64_t *rounds_out, void *stream) {
    IVX_REQUIRE(I && markers && C && strct && dz > 0 && dy > 0 && dx > 0, IVX_EINVAL, "sk_cost_levels: bad arguments");
    IVX_REQUIRE(dx % 64 == 0, IVX_EINVAL, "sk_cost_levels: needs dx %% 64 == 0");
    IVX_REQUIRE(mdtype == IVX_I16 || mdtype == IVX_I8, IVX_EINVAL, "sk_cost_levels: markers must be int16 or int8");
    IVX_REQUIRE(max_levels >= 1, IVX_EINVAL, "sk_cost_levels: max_levels");
    IVX_REQUIRE((((uintptr_t)I | (uintptr_t)markers) & 15) == 0, IVX_EINVAL, "sk_cost_levels: image and markers must be 16-byte aligned");
    ivx_flood_plan plan;
    plan.dz = dz; plan.dy = dy; plan.dx = dx; plan.wx = dx / 64;
    const int64_t s3[3] = {3, 3, 3};
    int rc = ivx_flood_strct_bits(strct, s3, &plan.strct_bits);
    if (rc) return rc;
    Tiles t;
    if ((rc = make_tiles(&plan, &t))) return rc;
    hipStream_t st = ivx::S(stream);
    const int64_t n = dz * dy * dx, nwords = n >> 6;
    const FScratch fs = make_fscratch(t);
    // workspace: R | candidate planes (the first `ahead` levels' made in one pass, then one at a time) | marker plane | flood scratch | snapshots
    const int ahead = max_levels < 4 ? max_levels : 4;
    const size_t pw = (size_t)nwords * 8, o_C = al256(pw), o_M = al256(o_C + (size_t)ahead * pw), o_S = al256(o_M + pw), o_P = al256(o_S + fs.total);
    void *mem;
    if ((rc = ivx::ws_get_s(ivx::WS_WSA, st, o_P + (size_t)max_levels * pw + 256, &mem))) return rc;
    unsigned long long *R = (unsigned long long *)mem, *cand0 = (unsigned long long *)((char *)mem + o_C);
    unsigned long long *mb = (unsigned long long *)((char *)mem + o_M);
    char *scr = (char *)mem + o_S, *snaps = (char *)mem + o_P;
    unsigned long long *d_count = (unsigned long long *)(scr + fs.off_status) + 2;
    if ((rc = ivx_dev_flood_clear(&plan, (uint64_t *)R, scr, stream))) return rc;
    const unsigned gw = (unsigned)ivx::cdiv(nwords, 256);
    uint8_t *dirty = (uint8_t *)(scr + fs.off_dirty0);
    int64_t rounds_total = 0, reached = 0;
    {
        const int64_t n8 = n >> 3;
        const unsigned g8 = (unsigned)ivx::cdiv(n8, 256);
        if (mdtype == IVX_I16) hipLaunchKernelGGL(k_ska_marker_bits<int16_t>, dim3(g8), dim3(256), 0, st, (const int16_t *)markers, n8, mb);
        else hipLaunchKernelGGL(k_ska_marker_bits<int8_t>, dim3(g8), dim3(256), 0, st, (const int8_t *)markers, n8, mb);
        IVX_LAUNCH_CHECK();
        switch (ahead) {
        case 1: hipLaunchKernelGGL(k_ska_cands<1>, dim3(g8), dim3(256), 0, st, I, n8, nwords, cand0); break;
        case 2: hipLaunchKernelGGL(k_ska_cands<2>, dim3(g8), dim3(256), 0, st, I, n8, nwords, cand0); break;
        case 3: hipLaunchKernelGGL(k_ska_cands<3>, dim3(g8), dim3(256), 0, st, I, n8, nwords, cand0); break;
        default: hipLaunchKernelGGL(k_ska_cands<4>, dim3(g8), dim3(256), 0, st, I, n8, nwords, cand0); break;
        }
        IVX_LAUNCH_CHECK();
    }
    int c = 0;
    for (; c < max_levels; c++) {
        unsigned long long *cand = c < ahead ? cand0 + (size_t)c * nwords : cand0;
        if (c >= ahead && (rc = ivx_dev_flood_candidates(&plan, IVX_U16, I, 0.0, (double)c, nullptr, 0, 0.0, (uint64_t *)cand, stream))) return rc;
        if (c > 0) {
            // "closed" tiles were closed for the previous level's candidate plane: this one has more candidates
            IVX_HIP(hipMemsetAsync(scr + fs.off_dirty0, 0, fs.off_cnt - fs.off_dirty0, st));
            hipLaunchKernelGGL(k_ska_frontier, dim3(gw), dim3(256), 0, st, t, cand, R, dirty);
            IVX_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_ska_seed_bits, dim3(gw), dim3(256), 0, st, t, mb, cand, R, dirty);
        IVX_LAUNCH_CHECK();
        ivx::ccl_invalidate(scr);
        int rounds = 0;
        if ((rc = flood_run_impl(&plan, (const uint64_t *)cand, false, (uint64_t *)R, scr, &rounds, stream))) return rc;
        rounds_total += rounds;
        IVX_HIP(hipMemcpyAsync(snaps + (size_t)c * pw, R, pw, hipMemcpyDeviceToDevice, st));
        IVX_HIP(hipMemsetAsync(d_count, 0, 8, st));
        hipLaunchKernelGGL(k_wsa_count, dim3(gw < 1024 ? gw : 1024), dim3(256), 0, st, R, nwords, d_count);
        IVX_LAUNCH_CHECK();
        uint32_t seq, got[2] = {0, 0};
        if ((rc = ivx::mailbox_publish(d_count, 2, st, &seq))) return rc;
        if ((rc = ivx::mailbox_wait(seq, st, got, 2))) return rc;
        reached = (int64_t)(((uint64_t)got[1] << 32) | got[0]);
        // enough is in -- or level 0 shows that this image has no plateau to speak of (a raw gradient: the bulk connects
        // dozens of levels up, and walking there level by level costs more than the relaxation it would save)
        if ((double)reached >= stop_frac * (double)n || (c == 0 && (double)reached < 0.05 * (double)n)) {
            c++;
            break;
        }
    }
    {
        const int64_t nchunks = n / 16, blocks = ivx::cdiv(nchunks, 256);
        hipLaunchKernelGGL(k_wsa_costs, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, (const uint16_t *)snaps,
                           nchunks, (int64_t)(pw / 2), c, C);
        IVX_LAUNCH_CHECK();
    }
    if (levels_done) *levels_done = c;
    if (reached_out) *reached_out = reached;
    if (rounds_out) *rounds_out = rounds_total;
    return IVX_OK;
}

extern "C" int ivx_dev_flood_run(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, void *scratch_,
                                 int *rounds, void *stream) {
    return flood_run_impl(p, cand, false, reached, scratch_, rounds, stream);
}
// clear + seed + run in one call: the flood of `seeds` over `cand` into a plane whose old contents are dead.  With at most
// 16 seeds and a standard structuring element the clearing and the seeding ride on the coarse pass (three launches before
// the rounds instead of five, no separate pass over the plane); otherwise the three calls run one after the other.
static int flood_grow_impl(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                           const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand, uint64_t *reached, void *scratch_,
                           int *rounds, void *stream, int resident) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    IVX_REQUIRE(ivx::dtype_size(dtype), IVX_EINVAL, "flood: unsupported dtype %d", dtype);
    for (int64_t n = 0; n < nseeds; n++) {
        const int64_t x = seeds_xyz[3 * n], y = seeds_xyz[3 * n + 1], z = seeds_xyz[3 * n + 2];
        IVX_REQUIRE(x >= 0 && y >= 0 && z >= 0 && x < t.dx && y < t.dy && z < t.dz, IVX_ERANGE,
                    "flood: seed (%lld,%lld,%lld) outside volume (%lld,%lld,%lld) [x,y,z]", (long long)x, (long long)y,
                    (long long)z, (long long)t.dx, (long long)t.dy, (long long)t.dz);
    }
    static const bool fused_on = [] {
        const char *e = getenv("IVX_FLOOD_FUSED");
        return !(e && e[0] == '0');
    }();
    if (!(fused_on && nseeds >= 1 && nseeds <= 16 && t.ntiles > 0 && flood_mode() == 1 && coarse_ok(t, false))) {
        if ((rc = ivx_dev_flood_clear(p, reached, scratch_, stream))) return rc;
        if ((rc = ivx_dev_flood_seed(p, dtype, data, t0, t1, seeds_xyz, nseeds, cand, reached, scratch_, stream))) return rc;
        return flood_run_impl(p, cand, false, reached, scratch_, rounds, stream, nullptr, false, resident);
    }
    ivx::ccl_invalidate(scratch_);
    Fresh f;
    f.dtype = dtype;
    f.data = data;
    f.t0 = t0;
    f.t1 = t1;
    f.sp.n = (int)nseeds;
    for (int64_t n = 0; n < nseeds; n++)
        for (int q = 0; q < 3; q++) f.sp.xyz[n][q] = seeds_xyz[3 * n + q];
    return flood_run_impl(p, cand, false, reached, scratch_, rounds, stream, &f, false, resident);
}
extern "C" int ivx_dev_flood_grow(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                                  const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand, uint64_t *reached, void *scratch_,
                                  int *rounds, void *stream) {
    return flood_grow_impl(p, dtype, data, t0, t1, seeds_xyz, nseeds, cand, reached, scratch_, rounds, stream, -1);
}
// ivx_dev_flood_grow that may return right behind a resident launch (k_flood_resident): *pending = 1 then, and the flood is
// complete for everything queued on `stream` after this call EXCEPT when the launch ends early (round cap, timed-out
// barrier) -- ivx_dev_flood_wait tells, finishes the flood, and the caller queues its dependent work again.  The point: the
// host is not in the loop while the flood runs, so the next stage is already in the queue when the last round ends.
extern "C" int ivx_dev_flood_grow_async(const ivx_flood_plan *p, int dtype, const void *data, double t0, double t1,
                                        const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand, uint64_t *reached,
                                        void *scratch_, int *rounds, int *pending, void *stream) {
    IVX_REQUIRE(pending, IVX_EINVAL, "flood_grow_async: NULL argument");
    *pending = 0;
    const int rc = flood_grow_impl(p, dtype, data, t0, t1, seeds_xyz, nseeds, cand, reached, scratch_, rounds, stream, 1);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_res_mu);
    *pending = g_res_pending.count(scratch_) ? 1 : 0;
    return IVX_OK;
}
extern "C" int ivx_dev_flood_wait(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, void *scratch_, int *rounds,
                                  int *late, void *stream) {
    return flood_wait_impl(p, cand, reached, scratch_, rounds, late, stream);
}

extern "C" int ivx_dev_flood_run_edges(const ivx_flood_plan *p, const uint64_t *edges, uint64_t *reached, void *scratch_,
                                       int *rounds, void *stream) {
    return flood_run_impl(p, edges, true, reached, scratch_, rounds, stream);
}

extern "C" int ivx_flood_edges_bytes(const ivx_flood_plan *p, size_t *nbytes) {
    size_t one;
    int rc = ivx_flood_bits_bytes(p, &one);
    if (rc) return rc;
    *nbytes = 6 * one;
    return IVX_OK;
}

extern "C" int ivx_dev_flood_edges_auto(const ivx_flood_plan *p, const int16_t *data, const uint8_t *out, float pfrac, int fill,
                                        uint64_t *edges, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    IVX_REQUIRE(data && out && edges, IVX_EINVAL, "flood_edges_auto: NULL argument");
    const int64_t total = t.dz * t.dy * t.wx * 8;
    if (!total) return IVX_OK;
    // the six planes are packed back to back, dz * dy * wx words each (ivx_flood_edges_bytes leaves room for that)
    hipLaunchKernelGGL(k_flood_edges_auto, dim3((unsigned)grid_for(total)), dim3(256), 0, ivx::S(stream), data, out, t, pfrac,
                       (uint8_t)fill, (uint8_t *)edges);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_flood_seed_forced(const ivx_flood_plan *p, const int64_t *seeds_xyz, int64_t nseeds, uint64_t *cand,
                                         uint64_t *reached, void *scratch_, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    for (int64_t n = 0; n < nseeds; n++) {
        const int64_t x = seeds_xyz[3 * n], y = seeds_xyz[3 * n + 1], z = seeds_xyz[3 * n + 2];
        IVX_REQUIRE(x >= 0 && y >= 0 && z >= 0 && x < t.dx && y < t.dy && z < t.dz, IVX_ERANGE,
                    "flood: seed (%lld,%lld,%lld) outside volume (%lld,%lld,%lld) [x,y,z]", (long long)x, (long long)y,
                    (long long)z, (long long)t.dx, (long long)t.dy, (long long)t.dz);
    }
    if (nseeds == 0) return IVX_OK;
    const FScratch s = make_fscratch(t);
    char *scr = (char *)scratch_;
    hipStream_t st = ivx::S(stream);
    ivx::ccl_invalidate(scratch_);
    int64_t *d_seeds = (int64_t *)(scr + s.off_seeds);
    for (int64_t b = 0; b < nseeds; b += (int64_t)SEED_CHUNK) {
        const int64_t m = nseeds - b < (int64_t)SEED_CHUNK ? nseeds - b : (int64_t)SEED_CHUNK;
        IVX_HIP(hipMemcpyAsync(d_seeds, seeds_xyz + 3 * b, (size_t)m * 24, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_flood_seed_forced, dim3((unsigned)ivx::cdiv(m, 256)), dim3(256), 0, st, t, d_seeds, m,
                           (unsigned long long *)cand, (unsigned long long *)reached, (uint8_t *)(scr + s.off_dirty0));
        IVX_LAUNCH_CHECK();
        IVX_HIP(hipStreamSynchronize(st)); // d_seeds is reused by the next chunk; seeds_xyz is pageable host memory
    }
    return IVX_OK;
}

extern "C" int ivx_dev_flood_arm_gate(const void *scratch, uint32_t *word, uint32_t value, uint32_t below_tiles) {
    IVX_REQUIRE(scratch && word, IVX_EINVAL, "flood_arm_gate: NULL argument");
    std::lock_guard<std::mutex> lk(g_gates_mu);
    g_gates[scratch] = GateArm{word, value, below_tiles};
    return IVX_OK;
}
// forget an arm that no flood consumed (the gate was opened by hand, or the buffers are about to be freed): the entry is
// keyed by the scratch ADDRESS, and a later allocation may get the same one
extern "C" int ivx_dev_flood_disarm_gate(const void *scratch) {
    std::lock_guard<std::mutex> lk(g_gates_mu);
    g_gates.erase(scratch);
    return IVX_OK;
}
extern "C" int ivx_dev_gate_open(uint32_t *word, uint32_t value, void *stream) {
    IVX_REQUIRE(word, IVX_EINVAL, "gate_open: NULL argument");
    hipLaunchKernelGGL(k_gate_set, dim3(1), dim3(64), 0, ivx::S(stream), word, value);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
extern "C" int ivx_dev_gate_wait(const uint32_t *word, uint32_t value, uint32_t timeout_us, void *stream) {
    IVX_REQUIRE(word, IVX_EINVAL, "gate_wait: NULL argument");
    hipLaunchKernelGGL(k_gate_wait, dim3(1), dim3(64), 0, ivx::S(stream), word, value, (unsigned long long)timeout_us * 100ull);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_flood_mark_slab(const ivx_flood_plan *p, void *scratch_, int64_t z0, int64_t z1, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    if (z0 < 0) z0 = 0;
    if (z1 > t.dz) z1 = t.dz;
    if (z1 <= z0 || t.ntiles == 0) return IVX_OK;
    const FScratch s = make_fscratch(t);
    const int64_t tz0 = z0 / TZ, tz1 = ivx::cdiv(z1, TZ);
    const int64_t n = (tz1 - tz0) * t.nty * t.wx;
    hipLaunchKernelGGL(k_flood_mark, dim3((unsigned)ivx::cdiv(n, 256)), dim3(256), 0, ivx::S(stream), t, tz0, tz1,
                       (uint8_t *)((char *)scratch_ + s.off_dirty0));
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_flood_apply(const ivx_flood_plan *p, const uint64_t *reached, int dtype, void *target,
                                   double fill, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    const int64_t total = t.dz * t.dy * t.wx * 8;
    if (!total) return IVX_OK;
    const int g = grid_for(total);
    hipStream_t st = ivx::S(stream);
    const uint8_t *r = (const uint8_t *)reached;
    if (dtype == IVX_U8 && t.dx % 64 == 0 && ((uintptr_t)target & 15) == 0) {
        const int64_t nchunks = t.dz * t.dy * t.dx / 16;
        const int64_t blocks = ivx::cdiv(nchunks, 1024);
        hipLaunchKernelGGL(k_flood_apply16, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st,
                           (const uint16_t *)reached, nchunks, (uint4 *)target, (uint8_t)fill);
        IVX_LAUNCH_CHECK();
        return IVX_OK;
    }
    switch (dtype) {
    case IVX_U8: hipLaunchKernelGGL(k_flood_apply<uint8_t>, dim3(g), dim3(256), 0, st, t, r, (uint8_t *)target, (uint8_t)fill); break;
    case IVX_I16: hipLaunchKernelGGL(k_flood_apply<int16_t>, dim3(g), dim3(256), 0, st, t, r, (int16_t *)target, (int16_t)fill); break;
    case IVX_U16: hipLaunchKernelGGL(k_flood_apply<uint16_t>, dim3(g), dim3(256), 0, st, t, r, (uint16_t *)target, (uint16_t)fill); break;
    case IVX_F64: hipLaunchKernelGGL(k_flood_apply<double>, dim3(g), dim3(256), 0, st, t, r, (double *)target, fill); break;
    default: ivx::set_error("flood: unsupported dtype %d", dtype); return IVX_EINVAL;
    }
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_flood_apply2(const ivx_flood_plan *p, const uint64_t *reached, uint8_t *out, int fill,
                                    uint8_t *mask, int select, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    const int64_t total = t.dz * t.dy * t.wx * 8;
    if (!total) return IVX_OK;
    hipLaunchKernelGGL(k_flood_apply2, dim3(grid_for(total)), dim3(256), 0, ivx::S(stream), t, (const uint8_t *)reached, out,
                       (uint8_t)fill, mask, (uint8_t)select);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_flood_count(const ivx_flood_plan *p, const uint64_t *reached, int64_t *count, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    *count = 0;
    const int64_t nw = t.dz * t.dy * t.wx;
    if (!nw) return IVX_OK;
    void *d_tot;
    if ((rc = ivx::ws_get_s(ivx::WS_SMALL, ivx::S(stream), 64, &d_tot))) return rc;
    hipStream_t st = ivx::S(stream);
    IVX_HIP(hipMemsetAsync(d_tot, 0, 8, st));
    hipLaunchKernelGGL(k_flood_count, dim3(grid_for(nw)), dim3(256), 0, st, (const unsigned long long *)reached, nw,
                       (unsigned long long *)d_tot);
    IVX_LAUNCH_CHECK();
    unsigned long long h = 0;
    IVX_HIP(hipMemcpyAsync(&h, d_tot, 8, hipMemcpyDeviceToHost, st));
    IVX_HIP(hipStreamSynchronize(st));
    *count = (int64_t)h;
    return IVX_OK;
}

// ---- multi-GPU halo: OR a neighbour's reached plane into slice z; counts words that gained bits -------------
// MARK: the tile of every word that gained bits is flagged dirty by the kernel itself (instead of the host marking the
// whole tile layer of z after reading the count back)
template <bool MARK>
__global__ __launch_bounds__(256) void k_flood_or_plane(unsigned long long *__restrict__ dst,
                                                        const unsigned long long *__restrict__ src,
                                                        const unsigned long long *__restrict__ cand, int64_t nwords,
                                                        unsigned int *__restrict__ changed, Tiles t, int64_t z,
                                                        uint8_t *__restrict__ dirty) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    const unsigned long long add = src[i] & cand[i] & ~dst[i];
    if (add) {
        dst[i] |= add;
        atomicAdd(changed, 1u);
        if (MARK) {
            const int64_t y = i / t.wx, txi = i - y * t.wx;
            mark_tile_nbhd(t, dirty, z / TZ, y / TY, txi);
        }
    }
}

__global__ __launch_bounds__(256) void k_bits_combine(unsigned long long *__restrict__ dst,
                                                      const unsigned long long *__restrict__ src, int64_t n, int op) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = op == 0 ? (dst[i] | src[i]) : (dst[i] & ~src[i]);
}
// whole-plane dst |= src (op 0) or dst &= ~src (op 1): keeps a derived inside plane in step with `mask[reached] = v`
extern "C" int ivx_dev_bits_combine(uint64_t *dst, const uint64_t *src, int64_t nwords, int op, void *stream) {
    IVX_REQUIRE(nwords >= 0 && (op == 0 || op == 1), IVX_EINVAL, "bits_combine: bad arguments");
    if (nwords == 0) return IVX_OK;
    const int64_t blocks = ivx::cdiv(nwords, 256);
    hipLaunchKernelGGL(k_bits_combine, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, ivx::S(stream),
                       (unsigned long long *)dst, (const unsigned long long *)src, nwords, op);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_flood_or_plane(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z,
                                      const uint64_t *plane, void *scratch_, int *changed, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    IVX_REQUIRE(z >= 0 && z < t.dz, IVX_ERANGE, "flood: plane %lld outside slab", (long long)z);
    const FScratch s = make_fscratch(t);
    const int64_t nw = t.dy * t.wx;
    *changed = 0;
    if (!nw) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    unsigned int *d_chg = (unsigned int *)((char *)scratch_ + s.off_status);
    IVX_HIP(hipMemsetAsync(d_chg, 0, 4, st));
    hipLaunchKernelGGL(k_flood_or_plane<false>, dim3((unsigned)ivx::cdiv(nw, 256)), dim3(256), 0, st,
                       (unsigned long long *)reached + z * nw, (const unsigned long long *)plane,
                       (const unsigned long long *)cand + z * nw, nw, d_chg, t, z, (uint8_t *)nullptr);
    IVX_LAUNCH_CHECK();
    unsigned int h = 0;
    IVX_HIP(hipMemcpyAsync(&h, d_chg, 4, hipMemcpyDeviceToHost, st));
    IVX_HIP(hipStreamSynchronize(st));
    *changed = (int)h;
    if (h) return ivx_dev_flood_mark_slab(p, scratch_, z - TZ, z + 1 + TZ, stream); // the layers that can see slice z
    return IVX_OK;
}

// The same, without any read-back: the number of words that gained bits is left in the caller's DEVICE word (overwritten),
// where the next exchange's all-reduce picks it up (ivx_comm_exchange_vote): a sharded region-growing round is
// enqueue-only up to its single host read.
static int or_planes_dev_impl(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z_a, const uint64_t *plane_a,
                              int64_t z_b, const uint64_t *plane_b, void *scratch_, uint32_t *changed_dev, void *stream, bool zero);
extern "C" int ivx_dev_flood_or_planes_dev(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z_a,
                                           const uint64_t *plane_a, int64_t z_b, const uint64_t *plane_b, void *scratch_,
                                           uint32_t *changed_dev, void *stream) {
    return or_planes_dev_impl(p, cand, reached, z_a, plane_a, z_b, plane_b, scratch_, changed_dev, stream, true);
}
// ... ADDED to the caller's device word (which ivx_dev_vote_read left at zero): no 4-byte fill in front of every round
extern "C" int ivx_dev_flood_or_planes_acc(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z_a,
                                           const uint64_t *plane_a, int64_t z_b, const uint64_t *plane_b, void *scratch_,
                                           uint32_t *changed_dev, void *stream) {
    return or_planes_dev_impl(p, cand, reached, z_a, plane_a, z_b, plane_b, scratch_, changed_dev, stream, false);
}
static int or_planes_dev_impl(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z_a, const uint64_t *plane_a,
                              int64_t z_b, const uint64_t *plane_b, void *scratch_, uint32_t *changed_dev, void *stream, bool zero) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    const FScratch s = make_fscratch(t);
    const int64_t nw = t.dy * t.wx;
    IVX_REQUIRE(changed_dev, IVX_EINVAL, "flood: null counter");
    hipStream_t st = ivx::S(stream);
    if (zero) IVX_HIP(hipMemsetAsync(changed_dev, 0, 4, st));
    if (!nw || (!plane_a && !plane_b)) return IVX_OK;
    uint8_t *dirty = (uint8_t *)((char *)scratch_ + s.off_dirty0);
    const int64_t zs[2] = {z_a, z_b};
    const uint64_t *pl[2] = {plane_a, plane_b};
    for (int q = 0; q < 2; q++) {
        if (!pl[q]) continue;
        IVX_REQUIRE(zs[q] >= 0 && zs[q] < t.dz, IVX_ERANGE, "flood: plane %lld outside slab", (long long)zs[q]);
        hipLaunchKernelGGL(k_flood_or_plane<true>, dim3((unsigned)ivx::cdiv(nw, 256)), dim3(256), 0, st,
                           (unsigned long long *)reached + zs[q] * nw, (const unsigned long long *)pl[q],
                           (const unsigned long long *)cand + zs[q] * nw, nw, (unsigned int *)changed_dev, t, zs[q], dirty);
        IVX_LAUNCH_CHECK();
    }
    return IVX_OK;
}

// Both halo planes of a slab in one call (either may be NULL): reached[z] |= plane & cand[z] for each, the tiles of the
// words that gained bits are marked dirty on the device, and ONE read-back returns the number of such words.
extern "C" int ivx_dev_flood_or_planes(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, int64_t z_a,
                                       const uint64_t *plane_a, int64_t z_b, const uint64_t *plane_b, void *scratch_,
                                       int *changed, void *stream) {
    Tiles t;
    int rc = make_tiles(p, &t);
    if (rc) return rc;
    const FScratch s = make_fscratch(t);
    const int64_t nw = t.dy * t.wx;
    *changed = 0;
    if (!nw || (!plane_a && !plane_b)) return IVX_OK;
    hipStream_t st = ivx::S(stream);
    unsigned int *d_chg = (unsigned int *)((char *)scratch_ + s.off_status);
    uint8_t *dirty = (uint8_t *)((char *)scratch_ + s.off_dirty0);
    IVX_HIP(hipMemsetAsync(d_chg, 0, 4, st));
    const int64_t zs[2] = {z_a, z_b};
    const uint64_t *pl[2] = {plane_a, plane_b};
    for (int q = 0; q < 2; q++) {
        if (!pl[q]) continue;
        IVX_REQUIRE(zs[q] >= 0 && zs[q] < t.dz, IVX_ERANGE, "flood: plane %lld outside slab", (long long)zs[q]);
        hipLaunchKernelGGL(k_flood_or_plane<true>, dim3((unsigned)ivx::cdiv(nw, 256)), dim3(256), 0, st,
                           (unsigned long long *)reached + zs[q] * nw, (const unsigned long long *)pl[q],
                           (const unsigned long long *)cand + zs[q] * nw, nw, d_chg, t, zs[q], dirty);
        IVX_LAUNCH_CHECK();
    }
    uint32_t seq, h = 0;
    if ((rc = ivx::mailbox_publish(d_chg, 1, st, &seq))) return rc;
    if ((rc = ivx::mailbox_wait(seq, st, &h, 1))) return rc;
    *changed = (int)h;
    return IVX_OK;
}

// ---- host forms -----------------------------------------------------------------------------------
static int flood_host(int dtype, void *data, const int64_t shape[3], const int64_t strides[3], const int64_t *seeds,
                      int64_t nseeds, double t0, double t1, double fill, const uint8_t *strct, const int64_t sshape[3],
                      uint8_t *out, const int64_t ostrides[3], int inplace) {
    using namespace ivx;
    const size_t isz = dtype_size(dtype);
    IVX_REQUIRE(isz, IVX_EINVAL, "floodfill: unsupported dtype %d", dtype);
    ivx_flood_plan plan;
    plan.dz = shape[0]; plan.dy = shape[1]; plan.dx = shape[2];
    plan.wx = cdiv(shape[2], 64);
    int rc;
    if ((rc = ivx_flood_strct_bits(strct, sshape, &plan.strct_bits))) return rc;
    // seed bounds are checked before anything is touched (the reference panics on an out-of-bounds seed)
    for (int64_t n = 0; n < nseeds; n++) {
        const int64_t x = seeds[3 * n], y = seeds[3 * n + 1], z = seeds[3 * n + 2];
        IVX_REQUIRE(x >= 0 && y >= 0 && z >= 0 && x < shape[2] && y < shape[1] && z < shape[0], IVX_ERANGE,
                    "floodfill: seed (%lld,%lld,%lld) outside volume", (long long)x, (long long)y, (long long)z);
    }
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0 || nseeds == 0) return IVX_OK;
    size_t bb, sb;
    if ((rc = ivx_flood_bits_bytes(&plan, &bb))) return rc;
    if ((rc = ivx_flood_scratch_bytes(&plan, &sb))) return rc;
    void *d_data, *d_out = nullptr, *d_cand, *d_reach, *d_scr;
    if ((rc = ws_get(WS_IN, n * isz, &d_data))) return rc;
    if ((rc = ws_get(WS_AUX0, bb, &d_cand))) return rc;
    if ((rc = ws_get(WS_AUX1, bb, &d_reach))) return rc;
    if ((rc = ws_get(WS_AUX2, sb, &d_scr))) return rc;
    if ((rc = upload_strided(d_data, data, shape, strides, isz, WS_IN))) return rc;
    if (!inplace) {
        if ((rc = ws_get(WS_OUT, n, &d_out))) return rc;
        if ((rc = upload_strided(d_out, out, shape, ostrides, 1, WS_OUT))) return rc;
    }
    if ((rc = ivx_dev_flood_candidates(&plan, dtype, d_data, t0, t1, (const uint8_t *)d_out, inplace ? 2 : 1, fill,
                                       (uint64_t *)d_cand, nullptr)))
        return rc;
    if ((rc = ivx_dev_flood_grow(&plan, dtype, d_data, t0, t1, seeds, nseeds, (uint64_t *)d_cand, (uint64_t *)d_reach, d_scr,
                                 nullptr, nullptr)))
        return rc;
    if (inplace) {
        if ((rc = ivx_dev_flood_apply(&plan, (const uint64_t *)d_reach, dtype, d_data, fill, nullptr))) return rc;
        IVX_HIP(hipDeviceSynchronize());
        return download_strided(data, shape, strides, d_data, isz, WS_IN);
    }
    if ((rc = ivx_dev_flood_apply(&plan, (const uint64_t *)d_reach, IVX_U8, d_out, fill, nullptr))) return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided(out, shape, ostrides, d_out, 1, WS_OUT);
}

extern "C" int ivx_floodfill_threshold(int dtype, const void *data, const int64_t shape[3], const int64_t strides[3],
                                       const int64_t *seeds_xyz, int64_t nseeds, double t0, double t1, int fill,
                                       const uint8_t *strct, const int64_t sshape[3], uint8_t *out,
                                       const int64_t out_strides[3]) {
    ivx::HostCallGuard host_guard__;
    return flood_host(dtype, (void *)data, shape, strides, seeds_xyz, nseeds, t0, t1, (double)(uint8_t)fill, strct,
                      sshape, out, out_strides, 0);
}

extern "C" int ivx_floodfill_threshold_inplace(int dtype, void *data, const int64_t shape[3], const int64_t strides[3],
                                               const int64_t *seeds_xyz, int64_t nseeds, double t0, double t1,
                                               double fill, const uint8_t *strct, const int64_t sshape[3]) {
    ivx::HostCallGuard host_guard__;
    return flood_host(dtype, data, shape, strides, seeds_xyz, nseeds, t0, t1, fill, strct, sshape, nullptr, strides, 1);
}

// floodfill_internal (floodfill.rs:5-49): 6-neighbour component of `data == v` around (x, y, z); the seed itself is
// filled and expanded whatever its value; voxels whose `out` byte already equals `fill` are barriers.
extern "C" int ivx_floodfill(int dtype, const void *data, const int64_t shape[3], const int64_t strides[3], int64_t x,
                             int64_t y, int64_t z, double v, int fill, uint8_t *out, const int64_t out_strides[3]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    const size_t isz = dtype_size(dtype);
    IVX_REQUIRE(isz, IVX_EINVAL, "floodfill: unsupported dtype %d", dtype);
    IVX_REQUIRE(x >= 0 && y >= 0 && z >= 0 && x < shape[2] && y < shape[1] && z < shape[0], IVX_ERANGE,
                "floodfill: seed (%lld,%lld,%lld) outside volume", (long long)x, (long long)y, (long long)z);
    ivx_flood_plan plan;
    plan.dz = shape[0]; plan.dy = shape[1]; plan.dx = shape[2];
    plan.wx = cdiv(shape[2], 64);
    plan.strct_bits = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 14) | (1u << 16) | (1u << 22); // 6 faces
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    size_t bb, sb;
    int rc;
    if ((rc = ivx_flood_bits_bytes(&plan, &bb))) return rc;
    if ((rc = ivx_flood_scratch_bytes(&plan, &sb))) return rc;
    void *d_data, *d_out, *d_cand, *d_reach, *d_scr;
    if ((rc = ws_get(WS_IN, n * isz, &d_data))) return rc;
    if ((rc = ws_get(WS_OUT, n, &d_out))) return rc;
    if ((rc = ws_get(WS_AUX0, bb, &d_cand))) return rc;
    if ((rc = ws_get(WS_AUX1, bb, &d_reach))) return rc;
    if ((rc = ws_get(WS_AUX2, sb, &d_scr))) return rc;
    if ((rc = upload_strided(d_data, data, shape, strides, isz, WS_IN))) return rc;
    if ((rc = upload_strided(d_out, out, shape, out_strides, 1, WS_OUT))) return rc;
    const double fl = (double)(uint8_t)fill;
    const int64_t seed[3] = {x, y, z};
    if ((rc = ivx_dev_flood_clear(&plan, (uint64_t *)d_reach, d_scr, nullptr))) return rc;
    if ((rc = ivx_dev_flood_candidates(&plan, dtype, d_data, v, v, (const uint8_t *)d_out, 1, fl, (uint64_t *)d_cand, nullptr)))
        return rc;
    if ((rc = ivx_dev_flood_seed_forced(&plan, seed, 1, (uint64_t *)d_cand, (uint64_t *)d_reach, d_scr, nullptr))) return rc;
    if ((rc = ivx_dev_flood_run(&plan, (const uint64_t *)d_cand, (uint64_t *)d_reach, d_scr, nullptr, nullptr))) return rc;
    if ((rc = ivx_dev_flood_apply(&plan, (const uint64_t *)d_reach, IVX_U8, d_out, fl, nullptr))) return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided(out, shape, out_strides, d_out, 1, WS_OUT);
}

// floodfill_auto_threshold (floodfill_py.rs:12-85): int16 data, uint8 out; every seed is filled and expanded
// unconditionally, a step from a voxel of value v reaches 6-neighbours in [ceil(v(1-p)), floor(v(1+p))] (f32, as i16).
extern "C" int ivx_floodfill_auto_threshold(const int16_t *data, const int64_t shape[3], const int64_t strides[3],
                                            const int64_t *seeds_xyz, int64_t nseeds, float pfrac, int fill, uint8_t *out,
                                            const int64_t out_strides[3]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    ivx_flood_plan plan;
    plan.dz = shape[0]; plan.dy = shape[1]; plan.dx = shape[2];
    plan.wx = cdiv(shape[2], 64);
    plan.strct_bits = (1u << 4) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 14) | (1u << 16) | (1u << 22);
    for (int64_t n = 0; n < nseeds; n++) {
        const int64_t x = seeds_xyz[3 * n], y = seeds_xyz[3 * n + 1], z = seeds_xyz[3 * n + 2];
        IVX_REQUIRE(x >= 0 && y >= 0 && z >= 0 && x < shape[2] && y < shape[1] && z < shape[0], IVX_ERANGE,
                    "floodfill_auto_threshold: seed (%lld,%lld,%lld) outside volume", (long long)x, (long long)y, (long long)z);
    }
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0 || nseeds == 0) return IVX_OK;
    size_t bb, eb, sb;
    int rc;
    if ((rc = ivx_flood_bits_bytes(&plan, &bb))) return rc;
    if ((rc = ivx_flood_edges_bytes(&plan, &eb))) return rc;
    if ((rc = ivx_flood_scratch_bytes(&plan, &sb))) return rc;
    void *d_data, *d_out, *d_edges, *d_reach, *d_scr;
    if ((rc = ws_get(WS_IN, n * 2, &d_data))) return rc;
    if ((rc = ws_get(WS_OUT, n, &d_out))) return rc;
    if ((rc = ws_get(WS_AUX0, eb, &d_edges))) return rc;
    if ((rc = ws_get(WS_AUX1, bb, &d_reach))) return rc;
    if ((rc = ws_get(WS_AUX2, sb, &d_scr))) return rc;
    if ((rc = upload_strided(d_data, data, shape, strides, 2, WS_IN))) return rc;
    if ((rc = upload_strided(d_out, out, shape, out_strides, 1, WS_OUT))) return rc;
    if ((rc = ivx_dev_flood_clear(&plan, (uint64_t *)d_reach, d_scr, nullptr))) return rc;
    if ((rc = ivx_dev_flood_edges_auto(&plan, (const int16_t *)d_data, (const uint8_t *)d_out, pfrac, fill, (uint64_t *)d_edges,
                                       nullptr)))
        return rc;
    if ((rc = ivx_dev_flood_seed_forced(&plan, seeds_xyz, nseeds, nullptr, (uint64_t *)d_reach, d_scr, nullptr))) return rc;
    if ((rc = ivx_dev_flood_run_edges(&plan, (const uint64_t *)d_edges, (uint64_t *)d_reach, d_scr, nullptr, nullptr))) return rc;
    if ((rc = ivx_dev_flood_apply(&plan, (const uint64_t *)d_reach, IVX_U8, d_out, (double)(uint8_t)fill, nullptr))) return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided(out, shape, out_strides, d_out, 1, WS_OUT);
}
