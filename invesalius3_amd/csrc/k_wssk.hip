// k_wssk.hip -- the marker flood of do_watershed's scikit-image branch ("Watershed", the GUI's default) on the GPU.
//
// Replaces skimage.segmentation.watershed(image, markers, bstruct) as called by
// invesalius/data/watershed_process.py:39,52 (3-D) and invesalius/data/styles.py:1958,1975 (one slice): no mask, no
// compactness, no watershed lines.  scikit-image's routine (_watershed_cy.pyx, watershed_raveled) is a serial binary-heap
// flood keyed (image value, age) with age = a global push counter, labels handed out at PUSH time.  What it computes can
// be said without the heap (oracle/ivx_oracle_wssk.c is the serial statement, tools/proto_ws_runs.py the prototype of
// this one):
//
//   cost   C(p) = min over paths from a marker of the largest image value ON the path (markers: C = I).  The heap pops
//          in non-decreasing C: level c = {C == c}.
//   GEN 0  of level c: the markers of value c (they carry age 0: before everything else, in raster order -- see TIES)
//          and the voxels of value c with a neighbour of lower C (queued before the level starts, ordered by the pop
//          time of the first such neighbour to pop).
//   STEPS  every other voxel of the level is reached from generation 0: a step INTO a voxel of value c costs one
//          generation (FIFO among equal values = breadth first), a step INTO a voxel of value < c costs nothing (the
//          heap drains a basin the moment it is reached, before the next voxel of value c pops).
//   TIME   T(p) = (G, R): G = generation counted across all levels, R = rank of p's generation-0 ancestor in its level's
//          sorted generation 0.  A voxel's label is the label of the neighbour with the smallest T.  The order INSIDE one
//          R never matters -- all those voxels carry one label, and a voxel contested by two labels is contested by two
//          different R -- which is also why scikit-image's neighbour order (an unstable argsort in 0.18) drops out.
//   TIES   marker voxels of equal image value all carry age 0 and leave scikit-image's heap in an order that depends on
//          the heap's array layout (every earlier push and pop).  That order is irrelevant whenever the tied markers
//          carry one label; otherwise it is not a function of the input a parallel machine can evaluate.  This kernel
//          takes tied markers in raster order and counts, in stats[6], the adjacent tied markers of different labels:
//          0 = identical to scikit-image by construction.
//
// Pipeline (one stream, no CPU arithmetic):
//   1. k_ws_relax<SK>  the cost map by chaotic relaxation over dirty 16x16x8 tiles (shared with the IFT flood);
//   2. k_sk_classify   generation-0 flags (EARLY: some neighbour lies below the level underneath -- the key is final one level
//                      early; LATE: all lower neighbours sit in the level right underneath), drained voxels (I < C) and their
//                      same-level drained neighbours; union-find makes every drained basin one set whose time stamp lives at
//                      its root; generation 0 bucketed by (level, late), the drained voxels by level;
//   3. per non-empty level, ascending:
//        generation 0 -- keys (marker: raster index; other: smallest T among lower-C neighbours), sorted, stamped (T, run
//        labels, first frontier) -- by our own kernels, no library: k_sk_keys -> k_sk_sort_chunks (bitonic in LDS) ->
//        k_sk_rank_pairs (pairwise ranks) | k_sk_merge_pass (lists of more than 128 chunks) -> k_sk_assign_ranked.  The EARLY
//        part of a level is sorted on a SIDE STREAM while the level underneath floods; behind that flood only the late part
//        is keyed and ONE launch stamps everything (k_sk_split_assign: late pairs ranked by brute force in LDS).
//        k_sk_gen0 (the whole of it in one launch behind a device-wide barrier) is an opt-in that measured slower.
//        Then the level's rounds in ONE resident launch (k_sk_level: a round boundary is a hand-over through a polled control
//        word; IVX_SK_PERSIST=0: k_sk_round launches in batches).  A
//      generation is at most two rounds: A -- the frontier stamps its unstamped neighbours of value c with the next
//      generation and offers its own stamp to the basins it touches (atomicMin at the root); B -- only if a basin was
//      stamped: the level's drained voxels whose basin carries this generation's stamp hand it, one generation later, to
//      their unstamped neighbours of value c.  Two shortcuts: a RUN of consecutive small levels is
//      taken by one workgroup in one launch (k_sk_levels_small: keys, LDS sort, stamps, generations behind barriers);
//      a basin-free level that is not small (the zero plateau of a windowed gradient) is relaxed tile-wise in LDS
//      (k_sk_plateau_relax: its stamps are a fixed point any relaxation order reaches).
//   4. k_sk_labels     label = label of the run in T's low word.
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <vector>

#include <cstring>

#include "ivx_internal.h"
#include "scan_u32.h"
#include "ws_tiles.h"

namespace {

constexpr unsigned long long TINF = ~0ull;
constexpr unsigned long long GEN1 = 1ull << 32;

// The frontier loop of a level runs without the host: every round is one launch of k_sk_round over a fixed grid; the
// last workgroup to finish turns the counters into the next round's input (phase B, or the next generation, or "level
// exhausted").  The host queues rounds in batches and reads {done, gen, rounds, n_in} once per batch.
struct SkState {
    uint32_t done;      // level exhausted: further rounds return at once   }
    uint32_t gen;       // G of the voxels on the current input list        } read by the host once per batch (mailbox)
    uint32_t rounds;    // rounds that did work (statistics)                }
    uint32_t n_in;      // entries of the current frontier                  }
    uint32_t phase;     // 0: the coming round is A (frontier), 1: B (basins stamped in this generation relay)
    uint32_t in_sel;    // which of the two lists is the frontier (the other collects the next generation)
    uint32_t n_next;    // voxels of the next generation so far
    uint32_t n_stamped; // basins stamped for the first time by this generation's round A
    uint32_t ticket;    // workgroups of the running round that have finished
    uint32_t mixed;     // adjacent tied markers with different labels (see TIES)
    uint32_t gens;      // generation steps (statistics)
    uint32_t brounds;   // B rounds (statistics)
    uint32_t gnext;     // first generation of the NEXT level (= gen + 1 when a level's loop ends): a chain of levels needs no host read
    // what a workgroup needs to decide whether it takes part in a launch, in ONE word (read with one atomic load):
    // bits 0..31 entries of the frontier, 32 phase, 33 which list, 34..63 sequence number of the launch it describes
    unsigned long long ctl;
    // k_sk_level's ticket (round 6): bits 48.. workgroups of the running round that have finished, 32..47 how many of them
    // stamped a basin for the first time, 0..31 list entries they appended -- the workgroup that closes the round learns all three
    // from the ONE returning atomic that tells it it is the last (before: ticket, then a dependent trip for the two counters)
    unsigned long long tick64;
    uint32_t pad1[14];
    // the resident launch (k_sk_level) polls its control word from every workgroup: a line of its own, away from the counters
    // the working workgroups add to
    unsigned long long pctl;
    uint32_t ticks[16]; // IVX_WS_TRACE: workgroup 0's time per part of a round, summed over the flood (wall_clock64 ticks of 10 ns)
    uint32_t joined;    // k_sk_level<.., LOCAL>: workgroups that found themselves on the chosen XCD (their ids: the order they joined in)
    uint32_t arrived;   //                        workgroups of the launch that have looked (all of them: `joined` is final then)
    uint32_t pad2[12];
};
static_assert(sizeof(SkState) == 256 && offsetof(SkState, pctl) == 128, "ctl is 8-byte aligned, pctl starts a 128-byte line");

__host__ __device__ inline unsigned long long sk_ctl(uint32_t seq, uint32_t in_sel, uint32_t phase, uint32_t n_in) {
    return ((unsigned long long)(seq & 0x3FFFFFFFu) << 34) | ((unsigned long long)(in_sel & 1u) << 33) |
           ((unsigned long long)(phase & 1u) << 32) | n_in;
}

// control word of the resident launch: bits 0..31 entries of the frontier, 32 phase, 33 which list, 34..53 generations begun
// since the level's first word (a solo stretch, below, begins several between two words), 54..63 sequence number of the word
// (mod PSEQ_MOD).  A workgroup without a share in a round may miss that round's word altogether: it takes the newest one.
constexpr uint32_t PSEQ_MOD = 0x3FFu, PSEQ_DONE = 0x3FFu, PGEN_MAX = 0xFFFFFu;
__host__ __device__ inline unsigned long long sk_pctl(uint32_t seq, uint32_t gens, uint32_t in_sel, uint32_t phase, uint32_t n_in) {
    return ((unsigned long long)(seq & 0x3FFu) << 54) | ((unsigned long long)(gens & 0xFFFFFu) << 34) |
           ((unsigned long long)(in_sel & 1u) << 33) | ((unsigned long long)(phase & 1u) << 32) | n_in;
}
struct SkLists {
    uint32_t *l[2];
};

struct SkKindPred {
    const uint8_t *kind;
    uint8_t which;
    __device__ bool operator()(int64_t p) const { return kind[p] == which; }
};
constexpr uint8_t KIND_GEN0 = 1, KIND_DRAINED = 2, KIND_GEN0_LATE = 3; // (late: see k_sk_classify)

// per voxel: generation 0 (markers, and voxels that sit AT their cost, I == C, next to a voxel of lower cost), or drained
// (I < C: part of a basin that is flooded the moment it is reached) with the mask of its drained neighbours of equal cost
template <int CONN, typename MT>
__global__ __launch_bounds__(256) void k_sk_classify(WsGeom g, const uint16_t *__restrict__ I, const uint16_t *__restrict__ C,
                                                     const MT *__restrict__ mk, uint8_t *__restrict__ kind, uint32_t *__restrict__ comp,
                                                     uint32_t *__restrict__ zmask, uint32_t *__restrict__ pmask, uint32_t *mbits,
                                                     const uint16_t *__restrict__ prevl) {
    __shared__ uint32_t s[NCELL];
    int z0, y0, x0;
    tile_origin(g, blockIdx.x, z0, y0, x0);
    load_tile<true>(g, z0, y0, x0, I, C, s);
    __syncthreads();
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    if (!(x0 + lx < g.w && y0 + ly < g.h)) return;
    const int nz = min(TZ, (int)(g.d - z0));
    for (int zz = 0; zz < nz; zz++) {
        const int ci = ((zz + 1) * BY + (ly + 1)) * BX + (lx + 1);
        const uint32_t cell = s[ci];
        const uint32_t c = cell >> 16, iv = cell & 0xFFFFu;
        const int64_t p = (int64_t)(z0 + zz) * g.hw + (int64_t)(y0 + ly) * g.w + (x0 + lx);
        const bool drained = c != CINF && iv < c; // (never a marker: those sit at their own value)
        uint32_t minlow = CINF; // smallest cost among the neighbours of lower cost
        uint32_t zm = 0, pm = 0; // neighbours of the same level: drained ones (zm), those at the level's value (pm)
#pragma unroll
        for (int k = 0; k < 27; k++) {
            if (!has_off<CONN>(g.smask, k)) continue;
            const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
            const uint32_t qv = s[ci + (dz * BY + dy) * BX + dx]; // cells outside the volume carry CINF
            if ((qv >> 16) < c) minlow = min(minlow, qv >> 16);
            zm |= ((qv >> 16) == c && (qv & 0xFFFFu) < c) ? 1u << k : 0u;
            pm |= ((qv >> 16) == c && (qv & 0xFFFFu) == c) ? 1u << k : 0u;
        }
        if (c == CINF) zm = pm = 0; // never reached: takes part in nothing
        const bool marker = mk[p] != 0, lower = minlow != CINF;
        // LATE: every neighbour of lower cost sits in the level right below (prevl[c]: the next lower level that has voxels), so
        // the key -- the smallest time stamp among them -- is known only when that level has been flooded.  Everybody else's
        // key is final one level earlier: their keys and sort run beside the level below (sk_run), and they all sort before
        // the late ones (a stamp of the level below is later than every earlier stamp).
        const bool late = !marker && lower && minlow == (uint32_t)prevl[c];
        kind[p] = (marker || (c != CINF && iv == c && lower)) ? (late ? KIND_GEN0_LATE : KIND_GEN0) : drained ? KIND_DRAINED : 0;
        // which levels hold markers (one bit per level; the host launches the tied-marker check for those only)
        if (marker && c != CINF && !((__hip_atomic_load(&mbits[c >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (c & 31u)) & 1u))
            atomicOr(&mbits[c >> 5], 1u << (c & 31u));
        comp[p] = drained ? (uint32_t)p : ENTRY;
        zmask[p] = zm; // (the union-find only follows it from drained voxels)
        pmask[p] = pm;
    }
}

__device__ __forceinline__ unsigned long long ld64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// keys of a level's generation 0: marker -> raster index (< 2^32 <= every T); other -> smallest T among lower-C neighbours
template <int CONN, typename MT>
__global__ __launch_bounds__(256) void k_sk_keys(WsGeom g, const uint16_t *__restrict__ C, const MT *__restrict__ mk,
                                                 const uint16_t *__restrict__ I, const uint32_t *__restrict__ comp,
                                                 const unsigned long long *tau, const uint32_t *__restrict__ elist,
                                                 unsigned long long *__restrict__ key, uint32_t *__restrict__ val, uint32_t cnt, uint32_t c) {
    // c: neighbours of cost below c count.  (The level itself -- or, for the voxels whose key is final one level earlier, the
    // level below: its stamps are being written while this runs and would lose against every earlier one anyway.)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    const uint32_t p = elist[i];
    unsigned long long K = p;
    if (mk[p] == 0) {
        K = TINF;
        const int64_t z = p / g.hw, r = p - z * g.hw, y = r / g.w, x = r - y * g.w;
#pragma unroll
        for (int k = 0; k < 27; k++) {
            if (!has_off<CONN>(g.smask, k)) continue;
            const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
            const int64_t Z = z + dz, Y = y + dy, X = x + dx;
            if ((uint64_t)X >= (uint64_t)g.w || (uint64_t)Y >= (uint64_t)g.h || (uint64_t)Z >= (uint64_t)g.d) continue;
            const int64_t q = (int64_t)p + dz * g.hw + dy * g.w + dx;
            const uint32_t qc = C[q];
            if (qc < c) K = min(K, ld64(&tau[(uint32_t)I[q] < qc ? comp[q] : (uint32_t)q])); // a drained voxel's stamp is its basin's
        }
    }
    key[i] = K;
    val[i] = p;
}

__device__ __forceinline__ bool sk_pair_less(unsigned long long ka, uint32_t va, unsigned long long kb, uint32_t vb) {
    return ka < kb || (ka == kb && va < vb);
}

// prevl[c] = the largest level below c that holds voxels (CINF: none), from the voxels-per-level histogram; one workgroup
__global__ __launch_bounds__(1024) void k_sk_prevlevel(const uint32_t *__restrict__ lhist, uint16_t *__restrict__ prevl) {
    __shared__ int s_last[1024];
    const int t = threadIdx.x; // levels 64 t .. 64 t + 63
    int last = -1;
    for (int q = 0; q < 64; q++)
        if (lhist[64 * t + q]) last = 64 * t + q;
    s_last[t] = last;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // running maximum
        const int v = t >= off ? s_last[t - off] : -1;
        __syncthreads();
        s_last[t] = max(s_last[t], v);
        __syncthreads();
    }
    int run = t ? s_last[t - 1] : -1;
    for (int q = 0; q < 64; q++) {
        prevl[64 * t + q] = run < 0 ? (uint16_t)CINF : (uint16_t)run;
        if (lhist[64 * t + q]) run = 64 * t + q;
    }
}

// generation 0 bucketed by (level, late) -- bucket 2 c holds the voxels of level c whose key is final one level early (markers
// among them), bucket 2 c + 1 the late ones: a level's list is one stretch, early part first -- and, in the same pass, the drained
// voxels by level (buckets 131072 + c: their stretches follow generation 0's in the same list).  k_ws_bucket's scheme (LDS
// counters for the low levels, their return values ARE the slots), one read of kind[] and C[] instead of two (round 6).
constexpr int BK3_D0 = 131072, BK3_N = 196608;
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_sk_bucket3(int64_t n, const uint16_t *__restrict__ C, const uint8_t *__restrict__ kind,
                                                    uint32_t *__restrict__ hist_or_cursor, uint32_t *__restrict__ elist,
                                                    const uint32_t *__restrict__ comp, uint32_t *__restrict__ droot, int lb) {
    // counters in LDS for the levels below lb (dynamic, 3 lb words: 2 lb for generation 0's buckets, lb for the drained; lb <= BK_LB,
    // sized by the host from the image's largest value -- with all 4096 levels, 48 KB a workgroup, three workgroups shared a compute
    // unit and cleared / flushed 12 288 counters per 16 384 voxels: 10 ms of a 1024^3 flood for 3 GB)
    extern __shared__ uint32_t sh[];
    uint32_t *shd = sh + 2 * lb;
    for (int i = threadIdx.x; i < 3 * lb; i += 256) sh[i] = 0;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * (256 * BK_CH);
    for (int pass = 0; pass < (SCATTER ? 2 : 1); pass++) {
        for (int j = 0; j < BK_CH; j++) {
            const int64_t p = b0 + (int64_t)j * 256 + threadIdx.x;
            const uint8_t kd = p < n ? kind[p] : (uint8_t)0;
            if (!kd) continue;
            const uint32_t c = C[p];
            uint32_t off;
            if (kd == KIND_DRAINED) {
                if (c < (uint32_t)lb) off = atomicAdd(&shd[c], 1u);
                else if (!SCATTER || pass == 1) off = atomicAdd(&hist_or_cursor[BK3_D0 + c], 1u);
                else continue;
            } else {
                const uint32_t bk = 2u * c + (kd == KIND_GEN0_LATE ? 1u : 0u);
                if (bk < (uint32_t)(2 * lb)) off = atomicAdd(&sh[bk], 1u);
                else if (!SCATTER || pass == 1) off = atomicAdd(&hist_or_cursor[bk], 1u);
                else continue;
            }
            if (SCATTER && pass == 1) {
                elist[off] = (uint32_t)p;
                if (kd == KIND_DRAINED && droot) droot[off] = comp[p]; // (its basin's root: a relay round reads it with the list)
            }
        }
        __syncthreads();
        if (pass == 0) {
            for (int i = threadIdx.x; i < 2 * lb; i += 256) {
                const uint32_t v = sh[i];
                if (v) {
                    const uint32_t base = atomicAdd(&hist_or_cursor[i], v);
                    if (SCATTER) sh[i] = base;
                }
            }
            for (int i = threadIdx.x; i < lb; i += 256) {
                const uint32_t v = shd[i];
                if (v) {
                    const uint32_t base = atomicAdd(&hist_or_cursor[BK3_D0 + i], v);
                    if (SCATTER) shd[i] = base;
                }
            }
            __syncthreads();
        }
    }
}

// sorted chunks + share planes -> one sorted list (the early part's sort runs beside the level below, on its own stream)
__global__ __launch_bounds__(256) void k_sk_scatter_sorted(const unsigned long long *__restrict__ key, const uint32_t *__restrict__ val,
                                                           const uint32_t *__restrict__ part, uint32_t shares, uint32_t ch,
                                                           unsigned long long *__restrict__ okey, uint32_t *__restrict__ oval, uint32_t cnt) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    uint32_t pos = i & (ch - 1u);
    for (uint32_t sh = 0; sh < shares; sh++) pos += part[(size_t)sh * cnt + i];
    okey[pos] = key[i];
    oval[pos] = val[i];
}

// A level's stamps when its generation 0 comes in two parts: the early part already sorted (position = index), the late part
// (keys from k_sk_keys, a few thousand at most) ranked here by brute force -- every late workgroup stages ALL late pairs in
// LDS, sixteen lanes share one pair's count (a sixteenth of the list each, eight probes in flight) -- and placed behind the early part.
constexpr uint32_t LATE_MAX = 12000, LATE_SPLIT = 16, LATE_PER_WG = 256 / LATE_SPLIT;
template <typename MT>
__global__ __launch_bounds__(256) void k_sk_split_assign(const unsigned long long *__restrict__ ekey, const uint32_t *__restrict__ evl, uint32_t cnt_e,
                                                         const unsigned long long *__restrict__ lkey, const uint32_t *__restrict__ lvl, uint32_t cnt_l,
                                                         const MT *__restrict__ mk, unsigned long long *tau, int32_t *runlabel,
                                                         uint32_t *__restrict__ front, uint32_t roff, uint32_t gbase, uint32_t seq, SkState *st) {
    extern __shared__ unsigned long long s_dyn[]; // cnt_l keys, then cnt_l voxels
    const uint32_t nwe = (cnt_e + 255u) / 256u, tid = threadIdx.x;
    if (gbase == 0) gbase = __hip_atomic_load(&st->gnext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && tid == 0) {
        const uint32_t cnt = cnt_e + cnt_l;
        st->done = 0; st->gen = gbase; st->n_in = cnt;
        st->phase = 0; st->in_sel = 0;
        st->n_next = 0; st->n_stamped = 0; st->ticket = 0; st->tick64 = 0; st->joined = 0; st->arrived = 0;
        st->ctl = sk_ctl(seq, 0, 0, cnt);
        st->pctl = sk_pctl(0, 0, 0, 0, cnt);
    }
    uint32_t p, pos;
    unsigned long long K;
    if (blockIdx.x < nwe) {
        const uint32_t i = blockIdx.x * 256 + tid;
        if (i >= cnt_e) return;
        p = evl[i]; K = ekey[i]; pos = i;
    } else {
        unsigned long long *s_k = s_dyn;
        uint32_t *s_v = (uint32_t *)(s_dyn + cnt_l);
        // (eight pairs per lane in flight: one round trip per 2 048 late pairs -- the plain loop made one per 256, six in a row for
        // a typical late part, in front of everything else this workgroup does)
        for (uint32_t i0 = tid; i0 < cnt_l; i0 += 8u * 256u) {
            unsigned long long rk[8];
            uint32_t rv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t i = i0 + (uint32_t)q * 256u;
                rk[q] = i < cnt_l ? lkey[i] : 0ull;
                rv[q] = i < cnt_l ? lvl[i] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t i = i0 + (uint32_t)q * 256u;
                if (i < cnt_l) {
                    s_k[i] = rk[q];
                    s_v[i] = rv[q];
                }
            }
        }
        __syncthreads();
        const uint32_t j = (blockIdx.x - nwe) * LATE_PER_WG + tid / LATE_SPLIT, part = tid % LATE_SPLIT;
        const bool act = j < cnt_l;
        K = act ? s_k[j] : TINF;
        p = act ? s_v[j] : 0xFFFFFFFFu;
        const uint32_t per = (cnt_l + LATE_SPLIT - 1u) / LATE_SPLIT, i0 = part * per, i1 = min(cnt_l, i0 + per);
        uint32_t below = 0;
#pragma unroll 8
        for (uint32_t i = i0; i < i1; i++) below += sk_pair_less(s_k[i], s_v[i], K, p) ? 1u : 0u;
#pragma unroll
        for (int o = 1; o < (int)LATE_SPLIT; o <<= 1) below += __shfl_xor(below, o, 64);
        if (!act || part != 0) return;
        pos = cnt_e + below;
    }
    const int m = (int)mk[p];
    const int32_t l = m ? (int32_t)m : (K == TINF ? 0 : runlabel[(uint32_t)(K & 0xFFFFFFFFull)]);
    runlabel[roff + pos] = l;
    tau[p] = ((unsigned long long)gbase << 32) | (unsigned long long)(roff + pos);
    front[pos] = p;
}

// ---- generation 0 of a level in ONE launch (k_sk_gen0): keys -> sorted -> stamps.
// A level's generation 0 is ~5 x 10^4 (512^3) to ~5 x 10^5 (1024^3) voxels, once per level, 170 levels per flood, every one
// on the flood's critical path.  As k_sk_keys + a library radix sort (one block sort + ~6 merge launches at this size) +
// k_sk_assign that was 10 dependent launches at the dispatch floor, ~80 us per level.  Here every workgroup computes the keys
// of one chunk, sorts (key, voxel) pairs in LDS (bitonic), publishes the sorted KEYS, and -- behind one device-wide barrier --
// ranks its own keys against every other chunk staged through LDS: position = own index + sum over the other chunks of the
// keys below (ties: the chunk with the lower index first), which is the position in the sorted whole.  Both lists are sorted,
// so a lane's consecutive keys continue where the previous one stopped (a gallop of ~2 probes instead of a 12-probe search).
// The order among equal keys is (chunk, voxel): it never matters -- equal keys carry one label (see TIME above) -- but it is
// a function of the level's list alone.  All workgroups must be resident (one per compute unit, 2^20 voxels at most; above:
// the launches of sk_sort_big below); a lost barrier ends the launch with ctl->fail = 1, never a hang.
constexpr int G0_T = 1024;
struct SkG0Ctl {
    uint32_t arrive[2]; // the two barriers of a launch
    uint32_t nmark;     // markers among the level's generation 0
    uint32_t exits;     // workgroups that have left: the last one clears the words above for the next launch
    uint32_t fail;      // a barrier timed out (sticky for the flood)
    uint32_t ticks[8];  // workgroup 0's time per phase, summed over the flood's launches (wall_clock64 ticks of 10 ns; IVX_WS_TRACE prints them)
    uint32_t pad[19];
};
static_assert(sizeof(SkG0Ctl) == 128, "one line");

// bitonic sort of n2 (a power of two) (key, voxel) pairs in LDS by all threads of the workgroup
__device__ __forceinline__ void sk_bitonic(unsigned long long *s_key, uint32_t *s_val, uint32_t n2) {
    for (uint32_t k2 = 2; k2 <= n2; k2 <<= 1)
        for (uint32_t j = k2 >> 1, lj = 31u - __clz(k2 >> 1); j > 0; j >>= 1, lj--) { // (shifts: j is not a compile-time constant)
            for (uint32_t t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
                const uint32_t lo = ((t >> lj) << (lj + 1)) | (t & (j - 1u)), hi = lo + j;
                const bool up = ((lo & k2) == 0);
                const unsigned long long ka = s_key[lo], kb = s_key[hi];
                const uint32_t va = s_val[lo], vb = s_val[hi];
                if (sk_pair_less(kb, vb, ka, va) == up) {
                    s_key[lo] = kb; s_key[hi] = ka;
                    s_val[lo] = vb; s_val[hi] = va;
                }
            }
            __syncthreads();
        }
}

// key of a generation-0 voxel (see k_sk_keys)
template <int CONN, typename MT>
__device__ __forceinline__ unsigned long long sk_key_of(const WsGeom &g, const uint16_t *__restrict__ C, const MT *__restrict__ mk,
                                                        const uint16_t *__restrict__ I, const uint32_t *__restrict__ comp,
                                                        const unsigned long long *tau, uint32_t p, uint32_t c, bool *marker) {
    *marker = mk[p] != 0;
    if (*marker) return p;
    unsigned long long K = TINF;
    const int64_t z = p / g.hw, r = p - z * g.hw, y = r / g.w, x = r - y * g.w;
#pragma unroll
    for (int k = 0; k < 27; k++) {
        if (!has_off<CONN>(g.smask, k)) continue;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const int64_t Z = z + dz, Y = y + dy, X = x + dx;
        if ((uint64_t)X >= (uint64_t)g.w || (uint64_t)Y >= (uint64_t)g.h || (uint64_t)Z >= (uint64_t)g.d) continue;
        const int64_t q = (int64_t)p + dz * g.hw + dy * g.w + dx;
        const uint32_t qc = C[q];
        if (qc < c) K = min(K, ld64(&tau[(uint32_t)I[q] < qc ? comp[q] : (uint32_t)q]));
    }
    return K;
}

constexpr uint32_t G0_SPIN_LIMIT = 1u << 21;

// device-wide barrier of a launch whose workgroups are all resident.  What a workgroup published before it (agent-scope,
// write-through stores) has been acknowledged when it arrives; what it reads after it, it reads with agent-scope loads.
__device__ __forceinline__ bool sk_g0_barrier(uint32_t *word, uint32_t nwg, uint32_t *fail, uint32_t *s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ok = 1;
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg; spins++) {
            if (spins > G0_SPIN_LIMIT) {
                __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        *s_flag = ok;
    }
    __syncthreads();
    const bool ok = *s_flag != 0;
    __syncthreads(); // (the flag word is reused by the next barrier)
    return ok;
}

// number of staged keys that sort before K (le: or equal to it), continuing from `from` keys already known to (gallop + bisection)
__device__ __forceinline__ uint32_t sk_count_from(const unsigned long long *sb, uint32_t ch, uint32_t from, unsigned long long K, bool le) {
    uint32_t lo = from, step = 1;
    while (lo + step <= ch) {
        const unsigned long long x = sb[lo + step - 1];
        if (!(le ? x <= K : x < K)) break;
        lo += step;
        step <<= 1;
    }
    for (step >>= 1; step; step >>= 1)
        if (lo + step <= ch) {
            const unsigned long long x = sb[lo + step - 1];
            if (le ? x <= K : x < K) lo += step;
        }
    return lo;
}

// the same count by bisection over the whole chunk (ch a power of two): a lane's first key
__device__ __forceinline__ uint32_t sk_count_bisect(const unsigned long long *sb, uint32_t ch, unsigned long long K, bool le) {
    uint32_t lo = 0;
    for (uint32_t step = ch >> 1; step; step >>= 1) {
        const unsigned long long x = sb[lo + step - 1];
        if (le ? x <= K : x < K) lo += step;
    }
    const unsigned long long x = sb[lo];
    return lo + ((le ? x <= K : x < K) ? 1u : 0u);
}

template <int CONN, typename MT, int IPT>
__global__ __launch_bounds__(G0_T) void k_sk_gen0(WsGeom g, const uint16_t *__restrict__ C, const MT *__restrict__ mk,
                                                  const uint16_t *__restrict__ I, const uint32_t *__restrict__ comp, unsigned long long *tau,
                                                  const uint32_t *__restrict__ elist, unsigned long long *ckey, int32_t *runlabel,
                                                  uint32_t *front, uint32_t cnt, uint32_t ch, uint32_t c, uint32_t roff, uint32_t gbase,
                                                  uint32_t seq, SkState *st, SkG0Ctl *ctl) {
    constexpr int CAP = G0_T * IPT;
    __shared__ unsigned long long s_key[CAP];
    __shared__ uint32_t s_val[CAP];
    __shared__ unsigned long long s_b[2][CAP];
    __shared__ uint32_t s_flag, s_nm;
    const uint32_t tid = threadIdx.x, chunk = blockIdx.x, nwg = gridDim.x;
    const uint32_t base = chunk * ch, nown = min(ch, cnt - base);
    // gbase 0: the level follows another one of the same chain on the device (nobody writes gnext during this launch)
    if (gbase == 0) gbase = __hip_atomic_load(&st->gnext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (chunk == 0 && tid == 0) { // the level's frontier loop starts from list 0, phase A (as k_sk_assign)
        st->done = 0; st->gen = gbase; st->n_in = cnt;
        st->phase = 0; st->in_sel = 0;
        st->n_next = 0; st->n_stamped = 0; st->ticket = 0; st->tick64 = 0; st->joined = 0; st->arrived = 0;
        st->ctl = sk_ctl(seq, 0, 0, cnt);
        st->pctl = sk_pctl(0, 0, 0, 0, cnt);
    }
    if (tid == 0) s_nm = 0;
    __syncthreads();
    unsigned long long tk = chunk == 0 && tid == 0 ? wall_clock64() : 0ull;
    auto tick = [&](int slot) {
        if (chunk == 0 && tid == 0) {
            const unsigned long long now = wall_clock64();
            atomicAdd(&ctl->ticks[slot], (uint32_t)(now - tk));
            tk = now;
        }
    };
    // ---- keys of this chunk, sorted in LDS
    uint32_t n2 = 1;
    while (n2 < nown) n2 <<= 1;
    uint32_t nm = 0;
    for (uint32_t i = tid; i < n2; i += G0_T) {
        unsigned long long K = TINF;
        uint32_t p = 0xFFFFFFFFu;
        if (i < nown) {
            p = elist[base + i];
            bool marker;
            K = sk_key_of<CONN, MT>(g, C, mk, I, comp, tau, p, c, &marker);
            nm += marker;
        }
        s_key[i] = K;
        s_val[i] = p;
    }
    if (nm) atomicAdd(&s_nm, nm);
    __syncthreads();
    if (tid == 0 && s_nm && nwg > 1) __hip_atomic_fetch_add(&ctl->nmark, s_nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tick(0);
    sk_bitonic(s_key, s_val, n2);
    tick(1);
    uint32_t pos[IPT]; // position of this lane's pairs (sorted positions IPT * tid + k) in the sorted whole
#pragma unroll
    for (int k = 0; k < IPT; k++) pos[k] = IPT * tid + k;
    if (nwg > 1) {
        for (uint32_t i = tid; i < nown; i += G0_T) __hip_atomic_store(&ckey[base + i], s_key[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!sk_g0_barrier(&ctl->arrive[0], nwg, &ctl->fail, &s_flag)) return;
        tick(2);
        // ---- rank against every other chunk (starting behind the own one: the chunks' readers spread over the chip)
        unsigned long long pre[IPT];
        auto fetch = [&](uint32_t oc) {
            const uint32_t ob = oc * ch, on = min(ch, cnt - ob);
#pragma unroll
            for (int e = 0; e < IPT; e++) {
                const uint32_t i = tid + e * G0_T;
                pre[e] = i < on ? __hip_atomic_load(&ckey[ob + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : TINF;
            }
        };
        unsigned long long own[IPT];
#pragma unroll
        for (int k = 0; k < IPT; k++) own[k] = IPT * tid + k < nown ? s_key[IPT * tid + k] : TINF;
        uint32_t oc = chunk + 1 == nwg ? 0 : chunk + 1;
        fetch(oc);
        int cur = 0;
        for (uint32_t s = 0; s + 1 < nwg; s++) {
            unsigned long long *sb = s_b[cur];
#pragma unroll
            for (int e = 0; e < IPT; e++)
                if (tid + e * G0_T < ch) sb[tid + e * G0_T] = pre[e];
            __syncthreads();
            const uint32_t this_oc = oc, on = min(ch, cnt - this_oc * ch);
            oc = oc + 1 == nwg ? 0 : oc + 1;
            if (s + 2 < nwg) fetch(oc); // in flight during the search
            const bool le = this_oc < chunk;
            uint32_t from = 0;
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                if (IPT * tid + k >= nown) break;
                from = k == 0 ? sk_count_bisect(sb, ch, own[k], le) : sk_count_from(sb, ch, from, own[k], le);
                pos[k] += min(from, on);
            }
            cur ^= 1;
        }
        tick(3);
    }
    // ---- stamps (G = gbase, R = roff + position), the runs' labels, the first frontier
    const uint32_t nmark = nwg > 1 ? __hip_atomic_load(&ctl->nmark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s_nm;
    const bool check = nmark >= 2;
    uint32_t mixed = 0;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint32_t q = IPT * tid + k;
        if (q >= nown) break;
        const uint32_t p = s_val[q];
        const unsigned long long K = s_key[q];
        const int m = (int)mk[p];
        // a run's label: its marker's, or the label of the run its parent belongs to (an earlier level: final)
        const int32_t l = m ? (int32_t)m : (K == TINF ? 0 : runlabel[(uint32_t)(K & 0xFFFFFFFFull)]);
        runlabel[roff + pos[k]] = l;
        tau[p] = ((unsigned long long)gbase << 32) | (unsigned long long)(roff + pos[k]);
        if (check && nwg > 1) __hip_atomic_store(&front[pos[k]], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else front[pos[k]] = p;
        // markers sort first, in raster order: adjacent tied markers of different labels (one chunk: the neighbour is at hand)
        if (check && nwg == 1 && m && q > 0 && (int)mk[s_val[q - 1]] != m) mixed++;
    }
    if (check && nwg > 1) {
        if (!sk_g0_barrier(&ctl->arrive[1], nwg, &ctl->fail, &s_flag)) return;
        for (uint32_t r = 1 + chunk * G0_T + tid; r < nmark; r += nwg * G0_T) {
            const uint32_t p0 = __hip_atomic_load(&front[r - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t p1 = __hip_atomic_load(&front[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mixed += mk[p0] != mk[p1];
        }
    }
    if (mixed) atomicAdd(&st->mixed, mixed);
    __syncthreads();
    tick(4);
    if (tid == 0 && nwg > 1) { // the last workgroup out clears the barrier words for the next launch
        if (__hip_atomic_fetch_add(&ctl->exits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
            ctl->arrive[0] = 0; ctl->arrive[1] = 0; ctl->nmark = 0; ctl->exits = 0;
        }
    }
}

// ---- levels of more than 2^20 generation-0 voxels (more chunks than workgroups that are resident at once): the same chunk
// sort, then merge passes as launches (merge path: every lane finds where its 8 outputs start in the two runs)
// 2048 pairs, 256 lanes, eight pairs per lane in registers: a trip through LDS serves up to three compare-exchange steps (the
// lane fetches the eight pairs whose indices differ in the three bits those steps pair up), 24 trips instead of 66 steps
// with a barrier each (the plain network, sk_bitonic, is LDS-throughput bound: 30 us per chunk).  One spare slot per eight
// keeps the strided trips at two-way bank conflicts.
__device__ __forceinline__ uint32_t sc_phys(uint32_t i) { return i + (i >> 3); }

template <int SC_T> // lanes: sorts up to 8 * SC_T pairs (chunks of 512 / 1024 / 2048)
__global__ __launch_bounds__(SC_T) void k_sk_sort_chunks(const unsigned long long *__restrict__ key, const uint32_t *__restrict__ val,
                                                         unsigned long long *__restrict__ okey, uint32_t *__restrict__ oval, uint32_t cnt, uint32_t ch) {
    constexpr int SC_N = 8 * SC_T, SC_LOG = SC_T == 256 ? 11 : SC_T == 128 ? 10 : 9;
    static_assert(SC_T == 64 || SC_T == 128 || SC_T == 256, "512, 1024 or 2048 pairs");
    __shared__ unsigned long long s_key[SC_N + SC_N / 8];
    __shared__ uint32_t s_val[SC_N + SC_N / 8];
    const uint32_t base = blockIdx.x * ch, nown = min(ch, cnt - base), tid = threadIdx.x;
    unsigned long long k[8];
    uint32_t v[8], idx[8];
    auto cex = [&](int a, int b, uint32_t k2) { // compare-exchange of registers a < b (indices idx[a] < idx[b]) at merge size k2
        const bool up = (idx[a] & k2) == 0;
        if (sk_pair_less(k[b], v[b], k[a], v[a]) == up) {
            const unsigned long long tk = k[a]; k[a] = k[b]; k[b] = tk;
            const uint32_t tv = v[a]; v[a] = v[b]; v[b] = tv;
        }
    };
    // the first trip starts from global memory: pairs 8 * tid .. 8 * tid + 7, merge sizes 2, 4, 8 entirely in registers
#pragma unroll
    for (int e = 0; e < 8; e++) {
        idx[e] = 8 * tid + e;
        k[e] = idx[e] < nown ? key[base + idx[e]] : TINF;
        v[e] = idx[e] < nown ? val[base + idx[e]] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int m = 1; m <= 3; m++)
#pragma unroll
        for (int bit = m - 1; bit >= 0; bit--)
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (!((e >> bit) & 1)) cex(e, e | (1 << bit), 1u << m);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        s_key[sc_phys(idx[e])] = k[e];
        s_val[sc_phys(idx[e])] = v[e];
    }
    __syncthreads();
    for (int m = 4; m <= SC_LOG; m++) {
        const uint32_t k2 = 1u << m;
        for (int top = m - 1; top >= 0; top -= 3) {
            const int B = top < 2 ? 2 : top, s0 = B - 2; // the trip's pairs differ in index bits s0 .. s0 + 2
            const uint32_t low = tid & ((1u << s0) - 1u), high = tid >> s0, b0 = (high << (B + 1)) | low;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                idx[e] = b0 | ((uint32_t)e << s0);
                k[e] = s_key[sc_phys(idx[e])];
                v[e] = s_val[sc_phys(idx[e])];
            }
            const int lo_bit = top - 2 < 0 ? 0 : top - 2;
            for (int bit = top; bit >= lo_bit; bit--) {
                const int eb = bit - s0; // (uniform)
                if (eb == 2) {
#pragma unroll
                    for (int e = 0; e < 4; e++) cex(e, e | 4, k2);
                } else if (eb == 1) {
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (!(e & 2)) cex(e, e | 2, k2);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) cex(e, e | 1, k2);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                s_key[sc_phys(idx[e])] = k[e];
                s_val[sc_phys(idx[e])] = v[e];
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < nown; i += SC_T) {
        okey[base + i] = s_key[sc_phys(i)];
        oval[base + i] = s_val[sc_phys(i)];
    }
}

// ---- the default for a level's generation 0 (up to 128 chunks): keys (k_sk_keys) -> sorted chunks (k_sk_sort_chunks) -> every
// key's position in the sorted whole = its position in its chunk + the keys below it in every other chunk (k_sk_rank_pairs:
// one workgroup per (chunk, share of the other chunks), the other chunk staged in LDS, a lane's consecutive keys galloping on
// from the previous one's count) -> stamps by position (k_sk_assign_ranked).  Four launches that use the whole chip instead of
// a block sort + ~6 merge passes at the dispatch floor; equal keys are ordered (chunk, voxel), which never matters (TIME).
template <int RP_E> // keys per lane (256 lanes: chunks of up to 256 * RP_E keys)
__global__ __launch_bounds__(256) void k_sk_rank_pairs(const unsigned long long *__restrict__ key, uint32_t *__restrict__ part, uint32_t cnt, uint32_t ch) {
    __shared__ unsigned long long sb[256 * RP_E + 1];
    const uint32_t a = blockIdx.x, nch = gridDim.x, base = a * ch, nown = min(ch, cnt - base), tid = threadIdx.x;
    // a lane's keys are 256 apart (a wave's lanes hold neighbouring keys: their probes fall into neighbouring slots), its eight
    // bisections run in lockstep, branch-free: twelve rounds of eight independent LDS reads
    unsigned long long own[RP_E];
    uint32_t acc[RP_E];
#pragma unroll
    for (int k = 0; k < RP_E; k++) {
        own[k] = tid + 256 * k < nown ? key[base + tid + 256 * k] : TINF;
        acc[k] = 0;
    }
    for (uint32_t b = blockIdx.y; b < nch; b += gridDim.y) {
        if (b == a) continue; // (uniform)
        const uint32_t ob = b * ch, on = min(ch, cnt - ob);
        __syncthreads(); // (the previous chunk has been searched)
        for (uint32_t i = tid; i <= ch; i += 256) sb[i] = i < on ? key[ob + i] : TINF;
        __syncthreads();
        const bool le = b < a;
        uint32_t lo[RP_E];
#pragma unroll
        for (int k = 0; k < RP_E; k++) lo[k] = 0;
        for (uint32_t step = ch >> 1; step; step >>= 1) {
#pragma unroll
            for (int k = 0; k < RP_E; k++) {
                const unsigned long long x = sb[lo[k] + step - 1];
                lo[k] += (le ? x <= own[k] : x < own[k]) ? step : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < RP_E; k++) {
            const unsigned long long x = sb[lo[k]];
            lo[k] += (le ? x <= own[k] : x < own[k]) ? 1u : 0u;
            acc[k] += min(lo[k], on);
        }
    }
    // this share's counts, one plane per share (summed by k_sk_assign_ranked: a million device-scope atomics per level were
    // the slow part of this kernel, 40 us)
    uint32_t *mine = part + (size_t)blockIdx.y * cnt + base;
#pragma unroll
    for (int k = 0; k < RP_E; k++)
        if (tid + 256 * k < nown) mine[tid + 256 * k] = acc[k];
}

// sorted chunks + positions -> time stamps (G = gbase, R = roff + position), the runs' labels, the first frontier
template <typename MT>
__global__ __launch_bounds__(256) void k_sk_assign_ranked(const unsigned long long *__restrict__ key, const uint32_t *__restrict__ val,
                                                          const uint32_t *__restrict__ part, uint32_t shares, uint32_t ch, const MT *__restrict__ mk,
                                                          unsigned long long *tau, int32_t *runlabel, uint32_t *__restrict__ front, uint32_t cnt,
                                                          uint32_t pos_off, uint32_t total, uint32_t roff, uint32_t gbase, uint32_t seq, SkState *st) {
    // (cnt entries of a level's `total`, placed from position pos_off on: a level's list may arrive in two sorted parts)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (gbase == 0) gbase = __hip_atomic_load(&st->gnext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i == 0 && pos_off == 0) {
        st->done = 0; st->gen = gbase; st->n_in = total;
        st->phase = 0; st->in_sel = 0;
        st->n_next = 0; st->n_stamped = 0; st->ticket = 0; st->tick64 = 0; st->joined = 0; st->arrived = 0;
        st->ctl = sk_ctl(seq, 0, 0, total);
        st->pctl = sk_pctl(0, 0, 0, 0, total);
    }
    if (i >= cnt) return;
    const uint32_t p = val[i];
    uint32_t pos = pos_off + (i & (ch - 1u)); // position inside its chunk + the keys below it in the other chunks (one plane per share)
    for (uint32_t sh = 0; sh < shares; sh++) pos += part[(size_t)sh * cnt + i];
    const unsigned long long K = key[i];
    const int m = (int)mk[p];
    const int32_t l = m ? (int32_t)m : (K == TINF ? 0 : runlabel[(uint32_t)(K & 0xFFFFFFFFull)]);
    runlabel[roff + pos] = l;
    tau[p] = ((unsigned long long)gbase << 32) | (unsigned long long)(roff + pos);
    front[pos] = p;
}

// adjacent tied markers of different labels (markers sort first, in raster order); launched for the levels that hold markers
template <typename MT>
__global__ __launch_bounds__(256) void k_sk_mixed(const uint32_t *__restrict__ front, const MT *__restrict__ mk, uint32_t cnt, SkState *st) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 || i >= cnt) return;
    const int m = (int)mk[front[i]];
    if (m && (int)mk[front[i - 1]] != m) atomicAdd(&st->mixed, 1u);
}

// One merge pass: runs of `run` pairs -> runs of 2 * run.  A workgroup produces MP_TILE consecutive outputs: two waves find,
// each with a 64-way search (64 probes per dependent round trip instead of one), where the tile's first and last diagonal cut
// the two runs (merge path; ties: run A first -- (key, voxel) pairs are unique anyway); the two input stretches go to LDS, every
// lane finds its own 8 outputs' cut there and merges them serially.
constexpr uint32_t MP_E = 8, MP_TILE = 256 * MP_E; // outputs per lane / per workgroup

// pairs of diagonal d's prefix that come from run A, by the wave: a0 / b0 the runs' starts, na / nb their lengths
__device__ __forceinline__ uint32_t mp_cut_wave(const unsigned long long *__restrict__ key, const uint32_t *__restrict__ val, uint32_t a0,
                                                uint32_t na, uint32_t b0, uint32_t nb, uint32_t d) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = d > nb ? d - nb : 0u, hi = min(d, na);
    while (lo < hi) { // (uniform)
        const uint32_t span = hi - lo, step = (span + 63u) / 64u, mid = lo + lane * step;
        bool p = false; // "A[mid] belongs to the prefix": true for a prefix of the lanes
        if (mid < hi) {
            const uint32_t j = d - 1u - mid;
            p = !sk_pair_less(key[b0 + j], val[b0 + j], key[a0 + mid], val[a0 + mid]);
        }
        const uint32_t t = (uint32_t)__popcll(__ballot(p));
        const uint32_t nlo = t ? lo + (t - 1u) * step + 1u : lo;
        const uint32_t mt = lo + t * step; // the first probe that failed (if any)
        hi = (t < 64u && mt < hi) ? mt : hi;
        lo = nlo;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_sk_merge_pass(const unsigned long long *__restrict__ key, const uint32_t *__restrict__ val,
                                                       unsigned long long *__restrict__ okey, uint32_t *__restrict__ oval, uint32_t cnt, uint32_t run,
                                                       uint32_t tile) {
    __shared__ unsigned long long s_k[MP_TILE];
    __shared__ uint32_t s_v[MP_TILE];
    __shared__ uint32_t s_cut[2];
    // tile = min(MP_TILE, 2 * run), both powers of two: a tile never straddles two pairs of runs
    const uint64_t o0 = (uint64_t)blockIdx.x * tile;
    const uint64_t pairb = o0 / (2ull * run) * (2ull * run);
    const uint32_t a0 = (uint32_t)pairb, na = cnt - pairb < run ? (uint32_t)(cnt - pairb) : run;
    const uint32_t b0 = a0 + na, nb = cnt - (pairb + na) < run ? (uint32_t)(cnt - (pairb + na)) : run;
    const uint32_t d0 = (uint32_t)(o0 - pairb), d1 = min(d0 + tile, na + nb), tid = threadIdx.x;
    if (tid < 128) { // waves 0 and 1
        const uint32_t c = mp_cut_wave(key, val, a0, na, b0, nb, tid < 64 ? d0 : d1);
        if ((tid & 63u) == 0) s_cut[tid >> 6] = c;
    }
    __syncthreads();
    const uint32_t i0 = s_cut[0], i1 = s_cut[1], j0 = d0 - i0, j1 = d1 - i1, la = i1 - i0, lb = j1 - j0; // la + lb == d1 - d0
    for (uint32_t x = tid; x < la + lb; x += 256) {
        const uint32_t src = x < la ? a0 + i0 + x : b0 + j0 + (x - la);
        s_k[x] = key[src];
        s_v[x] = val[src];
    }
    __syncthreads();
    // this lane's outputs: local diagonal dl = 8 * tid of the two staged stretches [0, la) and [la, la + lb)
    const uint32_t dl = tid * MP_E;
    if (dl >= la + lb) return;
    uint32_t lo = dl > lb ? dl - lb : 0u, hi = min(dl, la);
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1, j = dl - 1u - mid;
        if (!sk_pair_less(s_k[la + j], s_v[la + j], s_k[mid], s_v[mid])) lo = mid + 1u;
        else hi = mid;
    }
    uint32_t i = lo, j = dl - lo;
    const uint32_t nout = min(MP_E, la + lb - dl);
    const uint64_t ob = pairb + d0 + dl;
    for (uint32_t e = 0; e < nout; e++) {
        const bool take_b = i >= la || (j < lb && sk_pair_less(s_k[la + j], s_v[la + j], s_k[i], s_v[i]));
        const uint32_t src = take_b ? la + j : i;
        okey[ob + e] = s_k[src];
        oval[ob + e] = s_v[src];
        i += !take_b;
        j += take_b;
    }
}

// Appends to the next-generation list are staged per wave in LDS and handed to the global list with ONE atomic per wave
// and pass (a frontier of 10^5 voxels otherwise sends ~10^4 atomics to one counter word, one after the other).
constexpr int WB_CAP = 1024;
struct SkStage {
    volatile uint32_t buf[4][WB_CAP];
    volatile uint32_t n[4];
    uint32_t pushed, stamped; // what this workgroup appended / stamped in the running round (k_sk_level's ticket carries them)
};

// LOCAL (k_sk_level on ONE XCD, see there): what crosses workgroups is written with workgroup-scope operations -- performed in
// the XCD's L2, where the line STAYS -- instead of agent-scope ones (write-through to the fabric, line dropped); loads keep the
// agent-scope form (L1 bypassed, served by the L2) either way.
template <bool LOCAL> __device__ __forceinline__ uint32_t sk_add32(uint32_t *p, uint32_t v) {
    if constexpr (LOCAL) return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LOCAL> __device__ __forceinline__ unsigned long long sk_min64(unsigned long long *p, unsigned long long v) {
    if constexpr (LOCAL) return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LOCAL> __device__ __forceinline__ unsigned long long sk_add64(unsigned long long *p, unsigned long long v) {
    if constexpr (LOCAL) return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LOCAL> __device__ __forceinline__ void sk_store32(uint32_t *p, uint32_t v) {
    if constexpr (LOCAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LOCAL> __device__ __forceinline__ void sk_store64(unsigned long long *p, unsigned long long v) {
    if constexpr (LOCAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool LOCAL = false>
__device__ __forceinline__ void stage_push(bool want, uint32_t v, SkStage &sg, uint32_t *glist, uint32_t *gcnt) {
    const unsigned long long b = __ballot(want);
    if (!b) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int leader = __ffsll((long long)b) - 1;
    const uint32_t n = (uint32_t)__popcll(b), rank = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    if (lane == leader) {
        base = sg.n[wv];
        if (base + n <= WB_CAP) sg.n[wv] = base + n;
        else {
            base = 0x80000000u | sk_add32<LOCAL>(gcnt, n); // no room (rare): straight to the global list
            atomicAdd(&sg.pushed, n);
        }
    }
    base = __shfl(base, leader, 64);
    if (!want) return;
    if (base & 0x80000000u) sk_store32<LOCAL>(&glist[(base & 0x7FFFFFFFu) + rank], v);
    else sg.buf[wv][base + rank] = v;
}

// all lanes of the wave are here
template <bool LOCAL = false>
__device__ __forceinline__ void stage_flush(SkStage &sg, uint32_t *glist, uint32_t *gcnt) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = sg.n[wv];
    if (!n) return;
    uint32_t off = 0;
    if (lane == 0) {
        off = sk_add32<LOCAL>(gcnt, n);
        atomicAdd(&sg.pushed, n);
    }
    off = __shfl(off, 0, 64);
    // (agent-scope stores: inside the resident launch the next round's readers sit on other XCDs, behind other L2s -- unless LOCAL)
    for (uint32_t j = lane; j < n; j += 64) sk_store32<LOCAL>(&glist[off + j], sg.buf[wv][j]);
    if (lane == 0) sg.n[wv] = 0;
}

// stamp the unstamped neighbours of the level's value of voxel `v` (the set bits of its pmask) with `nt`, one generation
// after v's own stamp.  Lanes walk their own bits; the staged appends take whoever is there.
__device__ __forceinline__ void sk_offer_plateau(const WsGeom &g, unsigned long long *tau, uint32_t pm, uint32_t v, unsigned long long nt,
                                                 SkStage &sg, uint32_t *next, SkState *st) {
    while (pm) {
        const int k = __ffs(pm) - 1;
        pm &= pm - 1;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const uint32_t p = (uint32_t)((int64_t)v + dz * g.hw + dy * g.w + dx);
        bool push_n = false;
        if (nt < ld64(&tau[p])) push_n = atomicMin(&tau[p], nt) == TINF;
        stage_push(push_n, p, sg, next, &st->n_next);
    }
}

// One round.  Phase A: every frontier voxel (stamp t, generation G) stamps its unstamped neighbours of value c with
// t + one generation and offers t to the basins it touches (atomicMin at the basin's root; a basin is stamped by exactly
// one generation, and all its candidates arrive in this one launch).  Phase B: every drained voxel of the level whose basin
// carries a stamp of generation G relays it, one generation later, to its unstamped neighbours of value c.  Fixed grid,
// grid-stride over the list; the last workgroup out sets up the next round.
__global__ __launch_bounds__(256) void k_sk_round(WsGeom g, const uint32_t *__restrict__ pmask, const uint32_t *__restrict__ zmask,
                                                  const uint32_t *__restrict__ comp, unsigned long long *tau, SkLists L,
                                                  const uint32_t *__restrict__ dlist, uint32_t ndl, uint32_t seq, uint32_t per_wg, SkState *st) {
    // Only the workgroups that have list entries take part (and sign the ticket): a small frontier costs a handful of
    // atomics, not one per launched workgroup.  A workgroup without work may start after the last working one has already
    // set the state up for the NEXT launch -- hence one control word that names the launch it describes.
    if (__hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const unsigned long long ctl = __hip_atomic_load(&st->ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(ctl >> 34) != (seq & 0x3FFFFFFFu)) return;
    const uint32_t phase = (uint32_t)(ctl >> 32) & 1u, in_sel = (uint32_t)(ctl >> 33) & 1u, n_front = (uint32_t)ctl;
    const uint32_t n_in = phase ? ndl : n_front;
    const uint32_t nactive = min((uint32_t)gridDim.x, (n_in + per_wg - 1u) / per_wg); // (few signatures: several passes per workgroup)
    if (blockIdx.x >= nactive) return;
    const uint32_t gen = st->gen; // (a working workgroup runs before the update: the plain fields are this launch's)
    const uint32_t *__restrict__ in = phase ? dlist : L.l[in_sel];
    uint32_t *__restrict__ next = L.l[in_sel ^ 1u];
    const uint32_t stride = nactive * 256;
    __shared__ SkStage sg;
    if (threadIdx.x < 4) sg.n[threadIdx.x] = 0;
    __syncthreads();
    uint32_t stamped = 0;
    for (uint32_t i0 = blockIdx.x * 256; i0 < n_in; i0 += stride) { // (i0 is wave-uniform: the staged pushes stay convergent)
        const uint32_t i = i0 + threadIdx.x;
        bool act = i < n_in;
        const uint32_t v = act ? in[i] : 0u;
        if (phase == 0) {
            const unsigned long long t = act ? ld64(&tau[v]) : TINF;
            sk_offer_plateau(g, tau, act ? pmask[v] : 0u, v, t + GEN1, sg, next, st);
            uint32_t zm = (ndl && act) ? zmask[v] : 0u; // the basins this voxel touches
            uint32_t last = ENTRY;
            while (zm) {
                const int k = __ffs(zm) - 1;
                zm &= zm - 1;
                const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
                const uint32_t root = comp[(uint32_t)((int64_t)v + dz * g.hw + dy * g.w + dx)];
                if (root == last) continue; // (most neighbours of one voxel share a basin)
                last = root;
                if (t < ld64(&tau[root])) stamped += atomicMin(&tau[root], t) == TINF;
            }
        } else {
            unsigned long long tb = TINF;
            if (act) tb = ld64(&tau[comp[v]]);
            act = act && (uint32_t)(tb >> 32) == gen; // stamped by this generation (earlier ones have relayed already)
            sk_offer_plateau(g, tau, act ? pmask[v] : 0u, v, tb + GEN1, sg, next, st);
        }
        stage_flush(sg, next, &st->n_next);
    }
    if (stamped) atomicAdd(&st->n_stamped, stamped);
    __shared__ uint32_t s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(&st->ticket, 1u) == nactive - 1;
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    const uint32_t nst = __hip_atomic_load(&st->n_stamped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t nnx = __hip_atomic_load(&st->n_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st->ticket = 0;
    st->rounds += 1;
    if (phase == 0 && nst) { // basins were stamped: they relay before the generation advances
        st->phase = 1;
        st->n_stamped = 0;
        st->brounds += 1;
        __hip_atomic_store(&st->ctl, sk_ctl(seq + 1u, in_sel, 1, n_front), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (nnx) { // next generation
        st->phase = 0;
        st->n_in = nnx;
        st->n_next = 0;
        st->gen = gen + 1;
        st->gens += 1;
        st->in_sel = in_sel ^ 1u;
        __hip_atomic_store(&st->ctl, sk_ctl(seq + 1u, in_sel ^ 1u, 0, nnx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        st->done = 1;
    }
}

// The same rounds in ONE launch per level: the grid stays resident (one workgroup per compute unit) and a round boundary is
// a device-wide hand-over through the control word instead of a kernel boundary -- the last working workgroup of round r
// publishes the control word of round r + 1 (or SEQ_DONE), everybody else polls it.  A level of the 512^3 bench has ~6
// generations = ~10 rounds: as launches (queued 16 at a time) they cost 8.8 us each, working or not.
// Everything that crosses workgroups inside the launch -- stamps, counters, control word AND the list entries -- is an
// agent-scope atomic access (sc1 on gfx950: served at the point all XCDs share), so a round boundary needs no L2 write-back /
// invalidate: a workgroup waits for its own stores (s_waitcnt) and signs the ticket.  (With __threadfence() pairs instead,
// every round paid the write-back of eight L2s: 36 us per round, measured -- four times a launch.)
constexpr uint32_t SK_SPIN_LIMIT = 1u << 22; // polls of ~0.5 us: a lost hand-over ends the launch with done = 3, it never hangs

__device__ __forceinline__ uint32_t ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the offers of one voxel with every neighbour in flight at once (the loop over the set bits in sk_offer_plateau is a chain of
// dependent round trips, one per neighbour; inside the resident launch a round IS its chain of round trips)
template <int CONN, bool LOCAL = false>
__device__ __forceinline__ void sk_offer_plateau_wide(const WsGeom &g, unsigned long long *tau, uint32_t pm, uint32_t v, unsigned long long nt,
                                                      SkStage &sg, uint32_t *next, SkState *st) {
    unsigned long long old[27];
#pragma unroll
    for (int k = 0; k < 27; k++) {
        old[k] = 0ull;
        if (!has_off<CONN>(g.smask, k)) continue;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        if ((pm >> k) & 1u) old[k] = sk_min64<LOCAL>(&tau[(uint32_t)((int64_t)v + dz * g.hw + dy * g.w + dx)], nt);
    }
#pragma unroll
    for (int k = 0; k < 27; k++) {
        if (!has_off<CONN>(g.smask, k)) continue;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        if (!__ballot((pm >> k) & 1u)) continue; // (wave-uniform)
        stage_push<LOCAL>(old[k] == TINF, (uint32_t)((int64_t)v + dz * g.hw + dy * g.w + dx), sg, next, &st->n_next);
    }
}

#ifndef SK_POLL_SLEEP
#define SK_POLL_SLEEP 16 // (x 64 cycles between two polls of the control word)
#endif
#ifndef SK_TICKS
#define SK_TICKS 0 // 1: workgroup 0 times the parts of its rounds (build with -DSK_TICKS=1; tools/build_variant.py)
#endif
constexpr uint32_t SK_SOLO_MAX = 256; // list entries up to which ONE workgroup takes the round without a hand-over (one pass)

// one round's share of a workgroup: list entries wg * 256 + k * nactive * 256 (the whole workgroup is here)
template <int CONN, bool LOCAL>
__device__ __forceinline__ void sk_level_round(const WsGeom &g, const uint32_t *__restrict__ pmask, const uint32_t *__restrict__ zmask,
                                               const uint32_t *__restrict__ comp, unsigned long long *tau, const SkLists &L,
                                               const uint32_t *dlist, const uint32_t *__restrict__ droot, uint32_t ndl, SkState *st, SkStage &sg,
                                               uint32_t phase,
                                               uint32_t in_sel, uint32_t n_front, uint32_t gen, uint32_t wg, uint32_t nactive) {
    const uint32_t n_in = phase ? ndl : n_front;
    const uint32_t *in = phase ? dlist : L.l[in_sel];
    uint32_t *next = L.l[in_sel ^ 1u];
    const uint32_t stride = nactive * 256;
    if (threadIdx.x < 4) sg.n[threadIdx.x] = 0;
    if (threadIdx.x == 0) { sg.pushed = 0; sg.stamped = 0; }
    __syncthreads();
    const bool tr = SK_TICKS && wg == 0 && threadIdx.x == 0;
    unsigned long long tk = tr ? wall_clock64() : 0ull;
#define SK_TICK(slot)                                                          \
    if (tr) {                                                                  \
        const unsigned long long now_ = wall_clock64();                        \
        atomicAdd(&st->ticks[(slot) + (phase ? 6 : 0)], (uint32_t)(now_ - tk)); \
        tk = now_;                                                             \
    }
    uint32_t stamped = 0;
    for (uint32_t i0 = wg * 256; i0 < n_in; i0 += stride) {
        const uint32_t i = i0 + threadIdx.x;
        bool act = i < n_in;
        const uint32_t v = act ? (phase ? in[i] : ld32(&in[i])) : 0u; // (dlist is read-only; a frontier list was written in this launch)
        if (phase == 0) {
            const uint32_t pm = act ? pmask[v] : 0u, zm = (ndl && act) ? zmask[v] : 0u;
            const unsigned long long t = act ? ld64(&tau[v]) : TINF;
            if (SK_TICKS && tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SK_TICK(0) }
            uint32_t root[27]; // the basins this voxel touches: all their roots in flight, then all their stamps
#pragma unroll
            for (int k = 0; k < 27; k++) {
                root[k] = ENTRY;
                if (!has_off<CONN>(g.smask, k)) continue;
                const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
                if ((zm >> k) & 1u) root[k] = comp[(uint32_t)((int64_t)v + dz * g.hw + dy * g.w + dx)];
            }
            // the offers to the plateau neighbours AND to the basins' roots in flight together (round 6: the basins' atomics used to
            // be issued behind the plateau offers' returns -- a second dependent round trip per round, 2 us of its 12), then the
            // returns: pushes for the plateau voxels stamped first, the count of basins stamped first
            unsigned long long old[27], oldr[27];
            const unsigned long long nt = t + GEN1;
#pragma unroll
            for (int k = 0; k < 27; k++) {
                old[k] = 0ull;
                if (!has_off<CONN>(g.smask, k)) continue;
                const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
                if ((pm >> k) & 1u) old[k] = sk_min64<LOCAL>(&tau[(uint32_t)((int64_t)v + dz * g.hw + dy * g.w + dx)], nt);
            }
            uint32_t last = ENTRY;
#pragma unroll
            for (int k = 0; k < 27; k++) {
                oldr[k] = 0ull;
                if (!has_off<CONN>(g.smask, k)) continue;
                if (root[k] == ENTRY || root[k] == last) continue; // (most neighbours of one voxel share a basin)
                last = root[k];
                oldr[k] = sk_min64<LOCAL>(&tau[root[k]], t);
            }
#pragma unroll
            for (int k = 0; k < 27; k++) {
                if (!has_off<CONN>(g.smask, k)) continue;
                const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
                if (!__ballot((pm >> k) & 1u)) continue; // (wave-uniform)
                stage_push<LOCAL>(old[k] == TINF, (uint32_t)((int64_t)v + dz * g.hw + dy * g.w + dx), sg, next, &st->n_next);
            }
            if (SK_TICKS && tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SK_TICK(1) }
#pragma unroll
            for (int k = 0; k < 27; k++) {
                if (!has_off<CONN>(g.smask, k)) continue;
                stamped += oldr[k] == TINF;
            }
            if (SK_TICKS && tr) { SK_TICK(2) }
        } else {
            unsigned long long tb = TINF;
            if (act) tb = ld64(&tau[droot[i]]); // (the root travels with the list: no gather into comp[] on the relay's critical path)
            act = act && (uint32_t)(tb >> 32) == gen;
            if (SK_TICKS && tr) { SK_TICK(0) }
            sk_offer_plateau_wide<CONN, LOCAL>(g, tau, act ? pmask[v] : 0u, v, tb + GEN1, sg, next, st);
            if (SK_TICKS && tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SK_TICK(1) }
        }
        stage_flush<LOCAL>(sg, next, &st->n_next);
        if (SK_TICKS && tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SK_TICK(3) }
    }
    if (stamped) atomicAdd(&sg.stamped, stamped); // (LDS: the ticket carries it)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0); // this wave's stores and atomics have been acknowledged
    __syncthreads();
    SK_TICK(4)
#undef SK_TICK
}

// SOLO: most rounds of a level are tiny -- on the 512^3 bench 670 of the 990 frontier rounds have at most 1 024 voxels, and
// so have nearly all of the 780 basin relays -- and a hand-over (ticket, counters, control word, the others' poll) is four of
// a round's ten dependent round trips.  The workgroup that closes a round therefore keeps going ALONE while the coming round
// fits one workgroup (SK_SOLO_MAX entries): same lists, same counters, no ticket, no control word; the others keep polling
// for the word that ends the stretch -- a larger frontier, or the end of the level -- which tells them how many generations
// have begun meanwhile.
// Measured on the 512^3 bench: 64.8 -> 64.6 ms, i.e. nothing; neither did keeping the stretch's lists and counters in LDS (a
// wave's staged pushes as its share of the next frontier, the stamped-basins count an LDS word: 64.0 -> 63.3 ms, dropped).
// A round's time is its chain of dependent DRAM-latency accesses (~1.5 us each, agent-scope or not: pmask / zmask / comp are
// random reads of 0.5 GB arrays) -- voxel -> masks and stamp -> basin roots -> atomics, twice per generation with the relay
// -- and 1 742 generations times that chain is the level chain's floor; who runs the round hardly matters.
// LOCAL (round 6): the level's rounds on ONE XCD.  A round is a chain of dependent accesses to the stamps, the lists and the
// counters; at agent scope each of them is a trip over the fabric (the writer's L2 drops the line: the next reader misses too).
// The workgroups that find themselves on XCD 0 (HW_REG_XCC_ID, read at run time -- which workgroup lands where is not promised,
// so nobody assumes it) share one L2: between them a workgroup-scope atomic IS coherent (atomics execute in the L2, loads
// bypass the L1), and the level's working set stays in that L2 from round to round.  The others leave at once.  The launch
// is eight times as wide so that an eighth of it is enough; kernel boundaries make the result visible to everybody else.
// MEASURED (512^3, opt-in IVX_SK_LOCAL=1, same labels): level chain 44.6 - 46.8 ms against 39.1 with the rounds spread over all
// eight XCDs at agent scope -- 32 compute units and one L2's atomic unit serve the mid-sized rounds more slowly than the fabric
// trips cost.  Kept as an A/B, not the default.
template <int CONN, bool LOCAL>
__global__ __launch_bounds__(256) void k_sk_level(WsGeom g, const uint32_t *__restrict__ pmask, const uint32_t *__restrict__ zmask,
                                                  const uint32_t *__restrict__ comp, unsigned long long *tau, SkLists L,
                                                  const uint32_t *dlist, const uint32_t *__restrict__ droot, uint32_t ndl, uint32_t per_wg,
                                                  uint32_t solo_max, SkState *st) {
    __shared__ SkStage sg;
    __shared__ unsigned long long s_ctl;
    __shared__ uint32_t s_last, s_next[4], s_tot[2], s_keep; // s_next: phase, list, entries, 1 = level exhausted
    uint32_t my_wg = blockIdx.x, n_wg = gridDim.x;
    if (LOCAL) {
        __shared__ uint32_t s_id[2];
        if (threadIdx.x == 0) {
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const bool mine = (xcc & 7u) == 0u;
            uint32_t id = 0xFFFFFFFFu, n = 0;
            if (mine) id = __hip_atomic_fetch_add(&st->joined, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (returned: performed)
            __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine) {
                for (uint32_t spins = 0; ld32(&st->arrived) < gridDim.x; spins++) {
                    if (spins > SK_SPIN_LIMIT) { // (a workgroup of the launch never started: give up cleanly)
                        st32(&st->done, 3u);
                        id = 0xFFFFFFFFu;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                n = ld32(&st->joined);
            }
            s_id[0] = id;
            s_id[1] = n;
        }
        __syncthreads();
        if (s_id[0] == 0xFFFFFFFFu) return;
        my_wg = s_id[0];
        n_wg = s_id[1];
    }
    const uint32_t gen0 = ld32(&st->gen); // (k_sk_assign's launch wrote it)
    uint32_t want = 0;                    // sequence number of the next word this workgroup has not seen
    for (;;) {
        if (threadIdx.x == 0) {
            unsigned long long c;
            const unsigned long long tp0 = SK_TICKS && my_wg == 0 ? wall_clock64() : 0ull;
            for (uint32_t spins = 0;; spins++) {
                c = __hip_atomic_load(&st->pctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t sq = (uint32_t)(c >> 54);
                if (sq == PSEQ_DONE || (sq + PSEQ_MOD - want) % PSEQ_MOD < PSEQ_MOD / 2) break; // `want`, or a later one
                if (spins > SK_SPIN_LIMIT) {
                    st32(&st->done, 3u);
                    c = sk_pctl(PSEQ_DONE, 0, 0, 0, 0);
                    __hip_atomic_store(&st->pctl, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the others leave too)
                    break;
                }
                __builtin_amdgcn_s_sleep(SK_POLL_SLEEP);
            }
            if (SK_TICKS && my_wg == 0) atomicAdd(&st->ticks[12], (uint32_t)(wall_clock64() - tp0)); // waiting for the round's word
            s_ctl = c;
        }
        __syncthreads();
        const unsigned long long ctl = s_ctl;
        if ((uint32_t)(ctl >> 54) == PSEQ_DONE) return;
        const uint32_t seq = (uint32_t)(ctl >> 54);
        want = (seq + 1u) % PSEQ_MOD;
        uint32_t phase = (uint32_t)(ctl >> 32) & 1u, in_sel = (uint32_t)(ctl >> 33) & 1u, n_front = (uint32_t)ctl;
        uint32_t cum = (uint32_t)(ctl >> 34) & PGEN_MAX; // generations begun since the level's first word
        uint32_t gen = gen0 + cum;
        const uint32_t n_in = phase ? ndl : n_front;
        const uint32_t nactive = min(n_wg, (n_in + per_wg - 1u) / per_wg);
        if (my_wg < nactive) {
            sk_level_round<CONN, LOCAL>(g, pmask, zmask, comp, tau, L, dlist, droot, ndl, st, sg, phase, in_sel, n_front, gen, my_wg, nactive);
            const unsigned long long tt0 = SK_TICKS && my_wg == 0 && threadIdx.x == 0 ? wall_clock64() : 0ull;
            if (threadIdx.x == 0) {
                const uint32_t mine_p = sg.pushed, mine_s = sg.stamped ? 1u : 0u;
                const unsigned long long old = sk_add64<LOCAL>(&st->tick64, (1ull << 48) | ((unsigned long long)mine_s << 32) | mine_p);
                s_last = (uint32_t)(old >> 48) == nactive - 1;
                s_tot[0] = (uint32_t)old + mine_p;                       // list entries the whole round appended
                s_tot[1] = ((uint32_t)(old >> 32) & 0xFFFFu) + mine_s;   // workgroups of it that stamped a basin
            }
            __syncthreads();
            if (SK_TICKS && my_wg == 0 && threadIdx.x == 0) {
                atomicAdd(&st->ticks[13], (uint32_t)(wall_clock64() - tt0)); // the ticket
                atomicAdd(&st->ticks[14], 1u);                                // rounds workgroup 0 took part in
            }
            if (s_last) { // (the whole workgroup)
                // The ticket's low word counts the entries appended in this GENERATION (round A and, if there is one, its relay B append
                // to the same list): it is carried into a relay and cleared when the generation advances -- s_keep is what the word has
                // to hold when the next hand-over round starts.
                for (bool first = true;; first = false) {
                    if (threadIdx.x == 0) {
                        // (the round just closed: the ticket's totals; a round of the solo stretch: this workgroup's own counts)
                        const uint32_t nst = first ? s_tot[1] : (sg.stamped ? 1u : 0u);
                        const uint32_t nnx = first ? s_tot[0] : s_keep + sg.pushed;
                        sk_add32<LOCAL>(&st->rounds, 1u);
                        if (phase == 0 && nst) { // basins were stamped: they relay before the generation advances
                            sk_add32<LOCAL>(&st->brounds, 1u);
                            s_next[0] = 1u; s_next[1] = in_sel; s_next[2] = n_front; s_next[3] = 0u;
                            s_keep = nnx;
                        } else if (nnx) { // next generation
                            sk_store32<LOCAL>(&st->n_next, 0u);
                            sk_add32<LOCAL>(&st->gens, 1u);
                            s_next[0] = 0u; s_next[1] = in_sel ^ 1u; s_next[2] = nnx; s_next[3] = 0u;
                            s_keep = 0u;
                        } else {
                            s_next[3] = 1u;
                            s_keep = 0u;
                        }
                        __builtin_amdgcn_s_waitcnt(0);
                    }
                    __syncthreads();
                    if (s_next[3]) break;
                    const uint32_t nph = s_next[0], nsel = s_next[1], nfr = s_next[2];
                    __syncthreads(); // (s_next is rewritten after the next round)
                    const bool solo = (nph ? ndl : nfr) <= solo_max;
                    phase = nph; in_sel = nsel; n_front = nfr;
                    if (phase == 0) {
                        gen++;
                        cum++;
                    }
                    if (!solo) break;
                    sk_level_round<CONN, LOCAL>(g, pmask, zmask, comp, tau, L, dlist, droot, ndl, st, sg, phase, in_sel, n_front, gen, 0u, 1u);
                }
                if (threadIdx.x == 0) {
                    unsigned long long nctl;
                    if (s_next[3]) {
                        st32(&st->gen, gen); // the last generation used (what the host reads); the next level starts one later
                        st32(&st->gnext, gen + 1u);
                        st32(&st->done, 1u);
                        nctl = sk_pctl(PSEQ_DONE, 0, 0, 0, 0);
                    } else if (cum > PGEN_MAX) { // (more generations in one level than the word can count: give up cleanly)
                        st32(&st->done, 3u);
                        nctl = sk_pctl(PSEQ_DONE, 0, 0, 0, 0);
                    } else {
                        nctl = sk_pctl(want, cum, in_sel, phase, n_front);
                    }
                    sk_store64<LOCAL>(&st->tick64, (unsigned long long)s_keep); // (no arrivals yet; a relay continues its generation's count)
                    __builtin_amdgcn_s_waitcnt(0);
                    sk_store64<LOCAL>(&st->pctl, nctl);
                }
            }
        }
        __syncthreads(); // (s_ctl is rewritten by the next poll)
    }
}

// ---- a level without basins that is not small (the zero plateau of a windowed gradient image holds most of the volume): its
// breadth-first search has hundreds of generations, each a launch with a chain of dependent gathers.  The time stamps are
// the fixed point of  T(p) = min over neighbours q of the level (T(q) + one generation)  with generation 0 fixed, and any
// order of relaxation reaches it: so relax tiles staged in LDS to their local fixed point, dirty tiles only, like the
// cost map -- a tile crossing costs LDS round trips instead of launches.
constexpr unsigned long long TNM = TINF - 1ull; // staged cell that is not a voxel of the level (never offers, never accepts)

struct SkLevelPred { // every voxel that was reached
    __device__ bool operator()(int64_t) const { return true; }
};

// the tiles that stage a generation-0 voxel (its own, and those that see it in their halo) start dirty: the voxel itself
// never changes, so nobody else would wake the tile across the face
__global__ __launch_bounds__(256) void k_sk_mark_tiles(WsGeom g, const uint32_t *__restrict__ list, uint32_t cnt, uint8_t *__restrict__ dirty) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    const int64_t p = list[i], z = p / g.hw, r = p - z * g.hw, y = r / g.w, x = r - y * g.w;
    const int64_t tz = z / TZ, ty = y / TY, tx = x / TX;
    const int lz = (int)(z - tz * TZ), ly = (int)(y - ty * TY), lx = (int)(x - tx * TX);
    for (int dz = (lz == 0 ? -1 : 0); dz <= (lz == TZ - 1 ? 1 : 0); dz++)
        for (int dy = (ly == 0 ? -1 : 0); dy <= (ly == TY - 1 ? 1 : 0); dy++)
            for (int dx = (lx == 0 ? -1 : 0); dx <= (lx == TX - 1 ? 1 : 0); dx++) {
                const int64_t Z = tz + dz, Y = ty + dy, X = tx + dx;
                if (Z >= 0 && Z < g.ntz && Y >= 0 && Y < g.nty && X >= 0 && X < g.ntx) dirty[(Z * g.nty + Y) * g.ntx + X] = 1;
            }
}

template <int CONN>
__device__ __forceinline__ bool sk_plateau_eval(unsigned long long *s, uint32_t (*s_act)[TX], int lx, int ly, int zz, int nz, uint32_t smask) {
    const int ci = ((zz + 1) * BY + (ly + 1)) * BX + (lx + 1);
    const unsigned long long cur = s[ci];
    if (cur == TNM) return false;
    unsigned long long best = cur;
#pragma unroll
    for (int k = 0; k < 27; k++) {
        if (!has_off<CONN>(smask, k)) continue;
        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const unsigned long long q = s[ci + (dz * BY + dy) * BX + dx];
        if (q < TNM) best = min(best, q + GEN1); // (stamped voxels of the level only)
    }
    if (best >= cur) return false;
    s[ci] = best;
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

// The voxels AT the level's value (C == c and I == c) as a bit plane, linear voxel index = bit index: the tile-wise relaxation below
// stages 3 240 cells per tile visit, and reading C and I for each of them (68-byte rows out of 128-byte lines) was more than
// half of the bytes a visit moved -- 16 GB per flood, a burst of ~260 MB at the head of every round.  Lane = 8 voxels.
__global__ __launch_bounds__(256) void k_sk_level_plane(const uint16_t *__restrict__ C, const uint16_t *__restrict__ I, int64_t n, uint32_t c,
                                                        unsigned long long *__restrict__ P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; // chunk of 8 voxels; whole words: the grid covers ceil(n / 64) * 8 chunks
    uint32_t bits = 0;
    if (i * 8 + 8 <= n) {
        uint16_t vc[8], vi[8];
        *reinterpret_cast<uint4 *>(vc) = reinterpret_cast<const uint4 *>(C)[i];
        *reinterpret_cast<uint4 *>(vi) = reinterpret_cast<const uint4 *>(I)[i];
#pragma unroll
        for (int e = 0; e < 8; e++) bits |= ((uint32_t)vc[e] == c && (uint32_t)vi[e] == c ? 1u : 0u) << e;
    } else {
        for (int e = 0; e < 8; e++)
            if (i * 8 + e < n) bits |= ((uint32_t)C[i * 8 + e] == c && (uint32_t)I[i * 8 + e] == c ? 1u : 0u) << e;
    }
    const int lane = threadIdx.x & 63;
    unsigned long long w = (unsigned long long)bits << (8 * (lane & 7));
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)w, o, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(w >> 32), o, 64);
        w |= ((unsigned long long)hi << 32) | lo;
    }
    if ((lane & 7) == 0 && i * 8 < n) P[i >> 3] = w;
}

template <int CONN>
__global__ __launch_bounds__(256) void k_sk_plateau_relax(WsGeom g, const uint16_t *__restrict__ I, const uint16_t *__restrict__ C,
                                                          unsigned long long *tau, const uint32_t *__restrict__ list, uint8_t *dirty,
                                                          uint32_t c, SkState *st, const uint32_t *__restrict__ nlist, uint32_t offset,
                                                          const unsigned long long *__restrict__ P) {
    __shared__ unsigned long long s[NCELL];
    __shared__ uint32_t s_act[TY][TX];
    __shared__ uint32_t s_ev2, s_gmax;
    // (the grid is sized from the PREVIOUS round's list while the host is still reading this round's length: workgroups beyond
    // the list return at once, a longer list gets a second launch behind `offset`)
    if (blockIdx.x + offset >= *nlist) return;
    const int64_t tile = list[blockIdx.x + offset];
    int z0, y0, x0;
    tile_origin(g, tile, z0, y0, x0);
    constexpr int PER = (NCELL + 255) / 256;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int ce = threadIdx.x + q * 256;
        if (ce >= NCELL) continue;
        const int lx = ce % BX, ly = (ce / BX) % BY, lz = ce / (BX * BY);
        const int64_t zz = z0 + lz - 1, yy = y0 + ly - 1, xx = x0 + lx - 1;
        unsigned long long v = TNM;
        if ((uint64_t)xx < (uint64_t)g.w && (uint64_t)yy < (uint64_t)g.h && (uint64_t)zz < (uint64_t)g.d) {
            const int64_t Lx = zz * g.hw + yy * g.w + xx;
            const bool at = P ? (P[Lx >> 6] >> (Lx & 63)) & 1ull : (uint32_t)C[Lx] == c && (uint32_t)I[Lx] == c;
            if (at) v = tau[Lx];
        }
        s[ce] = v;
    }
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    s_act[ly][lx] = 0;
    if (threadIdx.x == 0) { s_ev2 = 0; s_gmax = 0; }
    __syncthreads();
    const bool col = x0 + lx < g.w && y0 + ly < g.h;
    const int nz = min(TZ, (int)(g.d - z0));
    uint32_t chg = 0;
    if (col)
        for (int zz = 0; zz < nz; zz++)
            if (sk_plateau_eval<CONN>(s, s_act, lx, ly, zz, nz, g.smask)) chg |= 1u << zz;
    __syncthreads();
    int it = 1;
    bool more = true;
    while (more && it < 4 * RELAX_ITCAP) {
        bool any = false;
        uint32_t a = col ? atomicExch(&s_act[ly][lx], 0u) : 0u;
        while (a) {
            const int zz = (it & 1) ? 31 - __clz(a) : __ffs(a) - 1;
            a &= ~(1u << zz);
            if (sk_plateau_eval<CONN>(s, s_act, lx, ly, zz, nz, g.smask)) {
                any = true;
                chg |= 1u << zz;
            }
        }
        more = __syncthreads_or(any);
        it++;
    }
    if (more && threadIdx.x == 0) dirty[tile] = 1; // iteration cap: come back
    uint32_t dirs = 0, gmax = 0;
    for (int zz = 0; zz < nz && chg; zz++) {
        if (!((chg >> zz) & 1u)) continue;
        const int z = z0 + zz, y = y0 + ly, x = x0 + lx;
        const unsigned long long t = s[((zz + 1) * BY + (ly + 1)) * BX + (lx + 1)];
        tau[(int64_t)z * g.hw + (int64_t)y * g.w + x] = t;
        gmax = max(gmax, (uint32_t)(t >> 32));
        const bool edge = lx == 0 || lx == TX - 1 || ly == 0 || ly == TY - 1 || zz == 0 || zz == TZ - 1;
        if (!edge) continue;
#pragma unroll
        for (int k = 0; k < 27; k++) {
            if (!has_off<CONN>(g.smask, k)) continue;
            const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
            const int Z = z + dz, Y = y + dy, X = x + dx;
            if ((unsigned)X >= (unsigned)g.w || (unsigned)Y >= (unsigned)g.h || (unsigned)Z >= (unsigned)g.d) continue;
            const int ex = X < x0 ? 0 : (X >= x0 + TX ? 2 : 1), ey = Y < y0 ? 0 : (Y >= y0 + TY ? 2 : 1), ez = Z < z0 ? 0 : (Z >= z0 + TZ ? 2 : 1);
            dirs |= 1u << (ez * 9 + ey * 3 + ex);
        }
    }
    if (dirs & ~(1u << 13)) atomicOr(&s_ev2, dirs);
    if (gmax) atomicMax(&s_gmax, gmax);
    __syncthreads();
    if (threadIdx.x < 27 && threadIdx.x != 13 && ((s_ev2 >> threadIdx.x) & 1u)) {
        const int k = threadIdx.x;
        const int tz = z0 / TZ + k / 9 - 1, ty = y0 / TY + (k / 3) % 3 - 1, tx = x0 / TX + k % 3 - 1;
        if (tz >= 0 && tz < g.ntz && ty >= 0 && ty < g.nty && tx >= 0 && tx < g.ntx) dirty[((int64_t)tz * g.nty + ty) * g.ntx + tx] = 1;
    }
    if (threadIdx.x == 0 && s_gmax > __hip_atomic_load(&st->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->gen, s_gmax);
}

// ---- small levels: everything a level needs in ONE workgroup and ONE launch, and a run of consecutive small levels in the
// same launch.  A slice of the GUI's 2-D mode, or the thin upper levels of a volume, are hundreds of levels of a few
// hundred voxels each: as separate launches (keys, sort, stamps, two per generation) they cost launch latency and nothing
// else.  Here: keys -> bitonic sort in LDS -> stamps -> the generations with workgroup barriers between their phases.
constexpr int SMALL_GEN0 = 4096;        // generation-0 voxels of a level that still sort in LDS
constexpr uint32_t SMALL_TOTAL = 32768; // voxels of a level one workgroup still walks

struct SkSmallArgs {
    const uint32_t *rec; // 32-byte key records of the run's entries (k_sk_small_recs), or nullptr
    uint32_t run_start;  // position in elist of the run's first entry
    const uint16_t *C, *I;
    const uint32_t *comp, *pmask, *zmask, *elist, *dlist;
    const uint32_t *hist, *cursor, *dhist, *dcursor; // counts and (after the scatter pass) segment ENDS per level
    unsigned long long *tau;
    int32_t *runlabel;
    uint32_t *list0, *list1;
    SkState *st;
};

// Key records for a run of small levels (6 neighbours): a generation-0 voxel's key is the smallest stamp among its neighbours of
// lower cost, read from tau[] at the neighbour or at its basin's root.  WHERE to read is settled before the flood starts -- list
// entry -> marker? -> the neighbours' costs -> their values and roots: three dependent round trips that the run's ONE workgroup made
// for every level (1.07 of its 2.2 ms at 512^3, measured by phase) -- so all the run's entries get a 32-byte record up front, by as
// many workgroups as it takes: { voxel, six places in tau (NONE: no such neighbour), marker }.  The run then reads the record and
// gathers the stamps: two round trips per level.
template <typename MT>
__global__ __launch_bounds__(256) void k_sk_small_recs(WsGeom g, const uint16_t *__restrict__ C, const MT *__restrict__ mk, const uint16_t *__restrict__ I,
                                                       const uint32_t *__restrict__ comp, const uint32_t *__restrict__ el, uint32_t n,
                                                       uint32_t *__restrict__ rec) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = el[i];
    uint32_t a[6] = {NONE, NONE, NONE, NONE, NONE, NONE};
    const bool marker = mk[p] != 0;
    if (!marker) {
        const uint32_t c = C[p];
        const int64_t z = p / g.hw, r = p - z * g.hw, y = r / g.w, x = r - y * g.w;
        const int64_t off[6] = {-g.hw, -g.w, -1, 1, g.w, g.hw};
        const bool ok[6] = {z > 0, y > 0, x > 0, x + 1 < g.w, y + 1 < g.h, z + 1 < g.d};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (!ok[k]) continue;
            const int64_t q = (int64_t)p + off[k];
            const uint32_t qc = C[q];
            if (qc < c) a[k] = (uint32_t)I[q] < qc ? comp[q] : (uint32_t)q; // (a drained voxel's stamp is its basin's)
        }
    }
    uint4 *r4 = reinterpret_cast<uint4 *>(rec + 8 * (size_t)i);
    r4[0] = make_uint4(p, a[0], a[1], a[2]);
    r4[1] = make_uint4(a[3], a[4], a[5], marker ? 1u : 0u);
}

// keys from such records (a level's late part: its records are made beside the level below, on the side stream)
__global__ __launch_bounds__(256) void k_sk_keys_rec(const uint32_t *__restrict__ rec, const unsigned long long *tau,
                                                     unsigned long long *__restrict__ key, uint32_t *__restrict__ val, uint32_t cnt) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    const uint4 *r4 = reinterpret_cast<const uint4 *>(rec + 8 * (size_t)i);
    const uint4 r0 = r4[0], r1 = r4[1];
    unsigned long long K = r0.x;
    if (!r1.w) {
        K = TINF;
        const uint32_t ad[6] = {r0.y, r0.z, r0.w, r1.x, r1.y, r1.z};
#pragma unroll
        for (int k = 0; k < 6; k++)
            if (ad[k] != NONE) K = min(K, ld64(&tau[ad[k]]));
    }
    key[i] = K;
    val[i] = r0.x;
}

template <typename MT>
__global__ __launch_bounds__(1024) void k_sk_levels_small(WsGeom g, SkSmallArgs a, const MT *__restrict__ mk, uint32_t c_lo, uint32_t c_hi,
                                                          uint32_t gbase, uint32_t roff) {
    __shared__ unsigned long long s_key[SMALL_GEN0];
    __shared__ uint32_t s_val[SMALL_GEN0];
    __shared__ uint32_t s_n_next, s_stamped, s_mixed;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_mixed = 0;
    for (uint32_t c = c_lo; c <= c_hi; c++) {
        const uint32_t cnt = a.hist[2 * c] + a.hist[2 * c + 1]; // (early + late part: one stretch of the list)
        if (!cnt) continue; // (uniform)
        const uint32_t ndl = a.dhist[c];
        const uint32_t *el = a.elist + (a.cursor[2 * c + 1] - cnt), *dl = a.dlist + (a.dcursor[c] - ndl);
        // keys
        uint32_t n2 = 1;
        while (n2 < cnt) n2 <<= 1;
        for (uint32_t i = tid; i < n2; i += 1024) {
            unsigned long long K = TINF;
            uint32_t p = 0;
            if (i < cnt && a.rec) { // (the places to read were found before the run: k_sk_small_recs)
                const uint4 *r4 = reinterpret_cast<const uint4 *>(a.rec + 8 * ((size_t)(el - a.elist) - a.run_start + i));
                const uint4 r0 = r4[0], r1 = r4[1];
                p = r0.x;
                K = p;
                if (!r1.w) {
                    K = TINF;
                    const uint32_t ad[6] = {r0.y, r0.z, r0.w, r1.x, r1.y, r1.z};
#pragma unroll
                    for (int k = 0; k < 6; k++)
                        if (ad[k] != NONE) K = min(K, ld64(&a.tau[ad[k]]));
                }
            } else if (i < cnt) {
                p = el[i];
                K = p;
                if (mk[p] == 0) {
                    K = TINF;
                    const int64_t z = p / g.hw, r = p - z * g.hw, y = r / g.w, x = r - y * g.w;
                    for (int k = 0; k < 27; k++) {
                        if (!((g.smask >> k) & 1u)) continue;
                        const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
                        const int64_t Z = z + dz, Y = y + dy, X = x + dx;
                        if ((uint64_t)X >= (uint64_t)g.w || (uint64_t)Y >= (uint64_t)g.h || (uint64_t)Z >= (uint64_t)g.d) continue;
                        const int64_t q = (int64_t)p + dz * g.hw + dy * g.w + dx;
                        const uint32_t qc = a.C[q];
                        if (qc < c) K = min(K, ld64(&a.tau[(uint32_t)a.I[q] < qc ? a.comp[q] : (uint32_t)q]));
                    }
                }
            }
            s_key[i] = K;
            s_val[i] = p;
        }
        __syncthreads();
        // bitonic sort of (key, voxel) pairs; padding keys (TINF) end up behind the real ones
        for (uint32_t k2 = 2; k2 <= n2; k2 <<= 1)
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < (n2 >> 1); t += 1024) {
                    const uint32_t lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                    const bool up = ((lo & k2) == 0);
                    const unsigned long long ka = s_key[lo], kb = s_key[hi];
                    if ((ka > kb) == up) {
                        s_key[lo] = kb; s_key[hi] = ka;
                        const uint32_t va = s_val[lo]; s_val[lo] = s_val[hi]; s_val[hi] = va;
                    }
                }
                __syncthreads();
            }
        // stamps, run labels, first frontier
        for (uint32_t i = tid; i < cnt; i += 1024) {
            const uint32_t p = s_val[i];
            const unsigned long long K = s_key[i];
            const int m = (int)mk[p];
            const int32_t l = m ? (int32_t)m : (K == TINF ? 0 : a.runlabel[(uint32_t)(K & 0xFFFFFFFFull)]);
            a.runlabel[roff + i] = l;
            a.tau[p] = ((unsigned long long)gbase << 32) | (unsigned long long)(roff + i);
            a.list0[i] = p;
            if (m && i > 0 && (int)mk[s_val[i - 1]] != m) atomicAdd(&s_mixed, 1u);
        }
        roff += cnt;
        // generations
        uint32_t n_in = cnt, gen = gbase;
        uint32_t *in = a.list0, *nx = a.list1;
        for (;;) {
            if (tid == 0) { s_n_next = 0; s_stamped = 0; }
            __syncthreads();
            for (uint32_t i = tid; i < n_in; i += 1024) { // phase A
                const uint32_t v = in[i];
                const unsigned long long t = ld64(&a.tau[v]), nt = t + GEN1;
                uint32_t pm = a.pmask[v];
                while (pm) {
                    const int k = __ffs(pm) - 1;
                    pm &= pm - 1;
                    const uint32_t p = (uint32_t)((int64_t)v + (k / 9 - 1) * g.hw + ((k / 3) % 3 - 1) * g.w + (k % 3 - 1));
                    if (nt < ld64(&a.tau[p]) && atomicMin(&a.tau[p], nt) == TINF) nx[atomicAdd(&s_n_next, 1u)] = p;
                }
                uint32_t zm = ndl ? a.zmask[v] : 0u, last = ENTRY;
                while (zm) {
                    const int k = __ffs(zm) - 1;
                    zm &= zm - 1;
                    const uint32_t root = a.comp[(uint32_t)((int64_t)v + (k / 9 - 1) * g.hw + ((k / 3) % 3 - 1) * g.w + (k % 3 - 1))];
                    if (root == last) continue;
                    last = root;
                    if (t < ld64(&a.tau[root]) && atomicMin(&a.tau[root], t) == TINF) s_stamped = 1;
                }
            }
            __syncthreads();
            if (s_stamped) { // phase B (uniform: read after the barrier)
                for (uint32_t i = tid; i < ndl; i += 1024) {
                    const uint32_t v = dl[i];
                    const unsigned long long tb = ld64(&a.tau[a.comp[v]]);
                    if ((uint32_t)(tb >> 32) != gen) continue;
                    const unsigned long long nt = tb + GEN1;
                    uint32_t pm = a.pmask[v];
                    while (pm) {
                        const int k = __ffs(pm) - 1;
                        pm &= pm - 1;
                        const uint32_t p = (uint32_t)((int64_t)v + (k / 9 - 1) * g.hw + ((k / 3) % 3 - 1) * g.w + (k % 3 - 1));
                        if (nt < ld64(&a.tau[p]) && atomicMin(&a.tau[p], nt) == TINF) nx[atomicAdd(&s_n_next, 1u)] = p;
                    }
                }
                __syncthreads();
            }
            const uint32_t nn = s_n_next;
            __syncthreads(); // (everybody has read the counters before they are cleared)
            if (!nn) break;
            n_in = nn;
            gen++;
            uint32_t *t2 = in; in = nx; nx = t2;
        }
        gbase = gen + 1;
    }
    if (tid == 0) {
        a.st->gen = gbase - 1; // what the host reads: the last generation used
        a.st->done = 1;
        if (s_mixed) atomicAdd(&a.st->mixed, s_mixed);
    }
}

template <typename MT>
__global__ __launch_bounds__(256) void k_sk_labels(int64_t n, const uint32_t *__restrict__ comp, const unsigned long long *__restrict__ tau,
                                                   const int32_t *__restrict__ runlabel, MT *__restrict__ out, int32_t *__restrict__ out32,
                                                   uint8_t *__restrict__ out8) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t cp = comp[p];
    const unsigned long long t = tau[cp == ENTRY ? (uint32_t)p : cp];
    const int32_t l = t == TINF ? 0 : runlabel[(uint32_t)(t & 0xFFFFFFFFull)];
    if (out) out[p] = (MT)l;
    if (out32) out32[p] = l;
    if (out8) out8[p] = (uint8_t)l;
}

__global__ void k_sk_fill64(unsigned long long *p, int64_t n, unsigned long long v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

constexpr uint64_t PLANE_CAP = 4u << 20; // counts in all share planes of one level (16 MB)
struct SkSortBufs { // working set of one generation-0 sort
    unsigned long long *key_a, *key_b;
    uint32_t *val_a, *val_b, *rank;
};
struct SkBufs {
    uint16_t *C, *prevl;
    uint8_t *kind, *dirty, *pending;
    unsigned long long *tau, *ekey[2];
    SkSortBufs sort[2];
    uint32_t *comp, *zmask, *pmask, *elist, *dlist, *droot, *lists[2], *eval[2], *hist, *cursor, *dhist, *dcursor, *lhist, *mbits, *bcount, *bsum, *tlist, *total;
    int32_t *runlabel;
    WsState *wst;
    SkState *st;
    SkG0Ctl *g0ctl;
    size_t bytes;
};

// fixed part, sized by the volume (slot WS_WSIFT)
static void sk_layout(const WsGeom &g, char *base, SkBufs *b) {
    size_t o = 0;
    auto take = [&](size_t n) { char *p = base ? base + o : nullptr; o += al(n); return p; };
    const int64_t nblk = cdiv(g.n, 2048);
    b->C = (uint16_t *)take((size_t)g.n * 2);
    b->kind = (uint8_t *)take((size_t)g.n);
    b->tau = (unsigned long long *)take((size_t)g.n * 8);
    b->comp = (uint32_t *)take((size_t)g.n * 4);
    b->zmask = (uint32_t *)take((size_t)g.n * 4);
    b->pmask = (uint32_t *)take((size_t)g.n * 4);
    b->dlist = nullptr; // (set once generation 0 has been counted: the drained voxels share elist, behind generation 0)
    b->droot = (uint32_t *)take((size_t)g.n * 4); // the drained entries' basin roots, slot for slot beside elist
    b->elist = (uint32_t *)take((size_t)g.n * 4);
    for (int i = 0; i < 2; i++) b->lists[i] = (uint32_t *)take((size_t)g.n * 4);
    b->hist = (uint32_t *)take((size_t)BK3_N * 4); // (level, late) buckets of generation 0, then the drained voxels' levels
    b->cursor = (uint32_t *)take((size_t)BK3_N * 4);
    b->prevl = (uint16_t *)take(65536 * 2);
    b->dhist = b->hist ? b->hist + BK3_D0 : nullptr;
    b->dcursor = b->cursor ? b->cursor + BK3_D0 : nullptr;
    b->lhist = (uint32_t *)take(65536 * 4);
    b->mbits = (uint32_t *)take(2048 * 4);
    b->bcount = (uint32_t *)take((size_t)(nblk + 1) * 4);
    b->bsum = (uint32_t *)take((size_t)(std::max<int64_t>(cdiv(nblk, 4096), 64) + 2) * 4);
    b->tlist = (uint32_t *)take((size_t)g.ntiles * 4);
    b->dirty = (uint8_t *)take((size_t)g.ntiles);
    b->pending = (uint8_t *)take((size_t)g.ntiles);
    b->wst = (WsState *)take(sizeof(WsState));
    b->st = (SkState *)take(sizeof(SkState));
    b->g0ctl = (SkG0Ctl *)take(sizeof(SkG0Ctl));
    b->total = (uint32_t *)take(256);
    b->bytes = o;
}

// the part sized once generation 0 has been counted (slot WS_WSSK)
static size_t sk_layout2(uint64_t ngen0, uint32_t maxcnt, bool split, char *base, SkBufs *b) {
    size_t o = 0;
    auto take = [&](size_t n) { char *p = base ? base + o : nullptr; o += al(n); return p; };
    b->runlabel = (int32_t *)take((size_t)(ngen0 + 1) * 4);
    for (int w = 0; w < (split ? 2 : 1); w++) { // [0]: the level chain's own stream; [1]: the early parts, sorted beside the level below
        SkSortBufs &sb = b->sort[w];
        sb.key_a = (unsigned long long *)take((size_t)maxcnt * 8 + 8);
        sb.key_b = (unsigned long long *)take((size_t)maxcnt * 8 + 8);
        sb.val_a = (uint32_t *)take((size_t)maxcnt * 4 + 8);
        sb.val_b = (uint32_t *)take((size_t)maxcnt * 4 + 8);
        sb.rank = (uint32_t *)take((size_t)std::min<uint64_t>((uint64_t)maxcnt * 512u, PLANE_CAP) * 4 + 8); // the share planes of k_sk_rank_pairs
    }
    for (int w = 0; w < 2; w++) { // two levels' sorted early parts (one being made while the other is read)
        b->ekey[w] = (unsigned long long *)take(split ? (size_t)maxcnt * 8 + 8 : 8);
        b->eval[w] = (uint32_t *)take(split ? (size_t)maxcnt * 4 + 8 : 8);
    }
    return o;
}

template <typename MT>
static int sk_run(const WsGeom &g, const uint16_t *I, const MT *mk, MT *out, int32_t *out32, uint8_t *out8, uint16_t *cost_out,
                  int64_t *stats, hipStream_t st) {
    const int conn = conn_of(g.smask);
    const int64_t nblk = cdiv(g.n, 2048);
    const int gl = (int)cdiv(g.n, 256);
    SkBufs b;
    sk_layout(g, nullptr, &b);
    void *mem = nullptr;
    IVX_REQUIRE(ws_get_s(WS_WSIFT, st, b.bytes, &mem) == IVX_OK, IVX_ENOMEM, "watershed: %zu bytes of scratch", b.bytes);
    sk_layout(g, (char *)mem, &b);

    WsTimer tm;
    tm.on = stats != nullptr;
    tm.mark(st);
    // ---- 1. costs ------------------------------------------------------------------------------------------
    IVX_HIP(hipMemsetAsync(b.wst, 0, sizeof(WsState), st));
    IVX_HIP(hipMemsetAsync(b.st, 0, sizeof(SkState), st));
    IVX_HIP(hipMemsetAsync(b.g0ctl, 0, sizeof(SkG0Ctl), st));
    IVX_HIP(hipMemsetAsync(b.dirty, 0, (size_t)g.ntiles, st));
    hipLaunchKernelGGL((k_ws_init<MT, true>), dim3((unsigned)nblk), dim3(256), 0, st, g, mk, I, b.C, b.dirty, b.bcount, b.wst);
    IVX_LAUNCH_CHECK();
    {
        const int rc = scan_u32_exclusive(b.bcount, nblk, b.bsum, b.total, st);
        if (rc != IVX_OK) return rc;
    }
    uint32_t M = 0;
    WsState hw;
    IVX_HIP(hipMemcpyAsync(&M, b.total, 4, hipMemcpyDeviceToHost, st));
    IVX_HIP(hipMemcpyAsync(&hw, b.wst, sizeof(hw), hipMemcpyDeviceToHost, st));
    IVX_HIP(hipStreamSynchronize(st));
    IVX_REQUIRE(!hw.overflow, IVX_EINVAL, "watershed: image value 65535 is reserved (the reference's gradient images stay far below)");
    if (M == 0) { // no marker: nothing is ever queued, every label stays 0
        if (out) IVX_HIP(hipMemsetAsync(out, 0, (size_t)g.n * sizeof(MT), st));
        if (out32) IVX_HIP(hipMemsetAsync(out32, 0, (size_t)g.n * 4, st));
        if (out8) IVX_HIP(hipMemsetAsync(out8, 0, (size_t)g.n, st));
        if (cost_out) IVX_HIP(hipMemsetAsync(cost_out, 0xFF, (size_t)g.n * 2, st));
        if (stats) memset(stats, 0, 16 * sizeof(int64_t));
        return IVX_OK;
    }
    const uint32_t nlev = std::min<uint32_t>(hw.imax + 2u, 65536u); // levels that can hold voxels: 0 .. the image's largest value
    int64_t rounds = 0, visits = 0;
    {
        // The levels that hold the bulk of the volume first, as ordinary region-growing floods (ivx_dev_sk_cost_levels):
        // the zero plateau of a windowed gradient -- ~95 % of the voxels -- is ONE flood; the relaxation keeps the rest.
        // IVX_SK_LEVELS = most levels (0: off), IVX_SK_LEVELS_FRAC, IVX_SK_LEVELS_MIN (voxels) as for the IFT branch.
        const char *e1 = getenv("IVX_SK_LEVELS"), *e2 = getenv("IVX_SK_LEVELS_FRAC"), *e3 = getenv("IVX_SK_LEVELS_MIN");
        const int lv_max = e1 ? atoi(e1) : 3;          // (512^3, windowed: cost map 9.7 ms without, 7.6 with one level, 5.7 with three)
        const double lv_frac = e2 ? atof(e2) : 0.99;
        const int64_t lv_min = e3 ? atoll(e3) : ((int64_t)1 << 21);
        if (lv_max > 0 && g.w % 64 == 0 && g.n >= lv_min && (((uintptr_t)I | (uintptr_t)mk) & 15) == 0) {
            uint8_t s27[27];
            for (int k = 0; k < 27; k++) s27[k] = (uint8_t)(k == 13 || ((g.smask >> k) & 1u));
            int levels_done = 0;
            int64_t lvox = 0, lrounds = 0;
            const int rc = ivx_dev_sk_cost_levels(I, sizeof(MT) == 2 ? IVX_I16 : IVX_I8, mk, g.d, g.h, g.w, s27, b.C, lv_max, lv_frac,
                                                  &levels_done, &lvox, &lrounds, st);
            if (rc != IVX_OK) return rc;
            // the levels' costs are final: only tiles that still hold a voxel without a cost have work (IVX_SK_RELAX_ALL=1: every tile
            // looks once, as in rounds 2 - 5 -- A/B)
            static const bool relax_all = []() { const char *e = getenv("IVX_SK_RELAX_ALL"); return e && e[0] == '1'; }();
            if (relax_all) {
                IVX_HIP(hipMemsetAsync(b.dirty, 1, (size_t)g.ntiles, st));
            } else {
                IVX_HIP(hipMemsetAsync(b.dirty, 0, (size_t)g.ntiles, st));
                hipLaunchKernelGGL(k_ws_mark_open_tiles, dim3((unsigned)cdiv(g.n / 8, 256)), dim3(256), 0, st, g, b.C, b.dirty);
                IVX_LAUNCH_CHECK();
            }
        }
        const int rc = ws_cost_rounds<true>(g, conn, I, b.C, b.tlist, b.dirty, b.pending, b.wst, st, &rounds, &visits);
        if (rc != IVX_OK) return rc;
    }
    if (cost_out) IVX_HIP(hipMemcpyAsync(cost_out, b.C, (size_t)g.n * 2, hipMemcpyDeviceToDevice, st));

    tm.mark(st);
    // ---- 2. generation 0 and the drained basins, bucketed by level ------------------------------------------
    const unsigned gbk = (unsigned)cdiv(g.n, 256 * BK_CH);
    std::vector<uint32_t> lhist(65536); // voxels per level (a level that holds much of the volume is relaxed tile-wise)
    IVX_HIP(hipMemsetAsync(b.lhist, 0, 65536 * 4, st));
    // LDS counters of the bucket passes: the levels that exist (no cost is above the image's largest value), whole 64s, BK_LB at most
    const int bk_lb = (int)std::min<uint32_t>(((hw.imax + 1u) + 63u) & ~63u, (uint32_t)BK_LB);
    hipLaunchKernelGGL((k_ws_bucket<SkLevelPred, false>), dim3(gbk), dim3(256), (size_t)bk_lb * 4, st, g.n, b.C, SkLevelPred{}, b.lhist, b.elist, bk_lb);
    IVX_LAUNCH_CHECK();
    IVX_HIP(hipMemcpyAsync(lhist.data(), b.lhist, (size_t)nlev * 4, hipMemcpyDeviceToHost, st)); // (only the levels the image has)
    hipLaunchKernelGGL(k_sk_prevlevel, dim3(1), dim3(1024), 0, st, b.lhist, b.prevl);
    IVX_LAUNCH_CHECK();
    IVX_HIP(hipMemsetAsync(b.mbits, 0, 2048 * 4, st));
    WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_sk_classify<CC, MT>), dim3((unsigned)g.ntiles), dim3(256), 0, st, g, I, b.C, mk, b.kind, b.comp,
                                              b.zmask, b.pmask, b.mbits, b.prevl));
    IVX_LAUNCH_CHECK();
    {
        const int rc = ws_zone_union(g, conn, b.zmask, b.comp, st);
        if (rc != IVX_OK) return rc;
    }
    IVX_HIP(hipMemsetAsync(b.hist, 0, (size_t)BK3_N * 4, st));
    std::vector<uint32_t> hist3(BK3_N), dhist(65536);
    { // generation 0 (a level's early part, then its late part) and, behind all of it, the drained voxels: ONE list, two passes
        hipLaunchKernelGGL(k_sk_bucket3<false>, dim3(gbk), dim3(256), (size_t)bk_lb * 12, st, g.n, b.C, b.kind, b.hist, b.elist, b.comp, (uint32_t *)nullptr, bk_lb);
        IVX_LAUNCH_CHECK();
        // (the buckets that can be used: 2 c + late and BK3_D0 + c for the levels the image has -- not 786 KB of zeros)
        IVX_HIP(hipMemcpyAsync(hist3.data(), b.hist, (size_t)nlev * 8, hipMemcpyDeviceToHost, st));
        IVX_HIP(hipMemcpyAsync(hist3.data() + BK3_D0, b.hist + BK3_D0, (size_t)nlev * 4, hipMemcpyDeviceToHost, st));
        IVX_HIP(hipMemcpyAsync(b.cursor, b.hist, (size_t)BK3_N * 4, hipMemcpyDeviceToDevice, st));
        const int rc = scan_u32_exclusive(b.cursor, BK3_N, b.bsum, b.total, st);
        if (rc != IVX_OK) return rc;
        hipLaunchKernelGGL(k_sk_bucket3<true>, dim3(gbk), dim3(256), (size_t)bk_lb * 12, st, g.n, b.C, b.kind, b.cursor, b.elist, b.comp, b.droot, bk_lb);
        IVX_LAUNCH_CHECK();
    }
    std::vector<uint32_t> mbits(2048); // levels that hold markers
    IVX_HIP(hipMemcpyAsync(mbits.data(), b.mbits, 2048 * 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_sk_fill64, dim3(2048), dim3(256), 0, st, b.tau, g.n, TINF);
    IVX_LAUNCH_CHECK();
    IVX_HIP(hipStreamSynchronize(st)); // the histograms are on the host now
    std::vector<uint32_t> hist(65536), hist_l(65536); // generation 0 per level: all of it, its late part
    uint64_t ngen0_all = 0;
    for (uint32_t c = 0; c < nlev; c++) {
        hist[c] = hist3[2 * c] + hist3[2 * c + 1];
        hist_l[c] = hist3[2 * c + 1];
        dhist[c] = hist3[BK3_D0 + c];
        ngen0_all += hist[c];
    }
    b.dlist = b.elist + ngen0_all; // (the drained voxels' stretches follow generation 0's in the one list)
    const uint32_t *droot_d = b.droot + ngen0_all;
    uint64_t ngen0 = 0;
    uint32_t maxcnt = 0;
    for (uint32_t c = 0; c < 65535; c++) { // (65535 = never reached: no generation 0 there)
        ngen0 += hist[c];
        maxcnt = std::max(maxcnt, hist[c]);
    }
    IVX_REQUIRE(ngen0 < 0xFFFFFFF0ull, IVX_EINVAL, "watershed: more than 2^32 queue entries");
    int ncu = 0;
    {
        int dev = 0;
        IVX_HIP(hipGetDevice(&dev));
        IVX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    // A level's generation 0 is keyed, sorted and stamped by our own kernels -- no library on this path:
    //   up to 128 chunks of 2048 voxels (the default): k_sk_keys -> k_sk_sort_chunks -> k_sk_rank_pairs -> k_sk_assign_ranked;
    //   more: k_sk_keys -> k_sk_sort_chunks -> k_sk_merge_pass x log2(chunks) -> k_sk_assign;
    //   IVX_SK_SORT=fused: everything in ONE launch with a device-wide barrier (k_sk_gen0; measured slower: DESIGN.md section 8),
    //   IVX_SK_SORT=merge: every level through the merge passes (tests, A/B);
    //   IVX_SK_CHUNK = n (a power of two up to 2048): the chunk length (tests: many chunks on small volumes).
    const char *ech = getenv("IVX_SK_CHUNK"), *eso2 = getenv("IVX_SK_SORT");
    uint32_t ch_env = ech ? (uint32_t)atoi(ech) : 0u;
    if (ch_env < 2 || ch_env > 2048 || (ch_env & (ch_env - 1))) ch_env = 0;
    const bool sort_merge = eso2 && eso2[0] == 'm', sort_fused = eso2 && eso2[0] == 'f';
    const uint32_t g0_wgs = (uint32_t)std::max(ncu, 8);
    // chunks of 1024 voxels (the chunk sort is bound by one compute unit's instruction rate: 15 us for 1024 pairs, 31 for 2048)
    // until the pairwise ranks outweigh that -- their work grows with voxels x chunks; IVX_SK_CHUNK_SWITCH, _PAIR_CHUNKS, _PAIR_WGS: A/B
    const char *e_sw = getenv("IVX_SK_CHUNK_SWITCH"), *e_pc = getenv("IVX_SK_PAIR_CHUNKS"), *e_pw = getenv("IVX_SK_PAIR_WGS");
    const uint32_t ch_switch = e_sw ? (uint32_t)atoll(e_sw) : (1u << 17);
    const int64_t PAIR_CHUNKS = e_pc ? atoll(e_pc) : 128, PAIR_WGS = e_pw ? std::max<int64_t>(atoll(e_pw), 1) : 2048;
    auto chunk_len = [&](uint32_t cnt) -> uint32_t { return ch_env ? ch_env : cnt <= ch_switch ? 1024u : 2048u; };
    auto g0_chunk = [&](uint32_t cnt) -> uint32_t { // chunk length of the one-launch path, 0: not for this level
        if (!sort_fused) return 0u;
        if (ch_env) return cdiv((int64_t)cnt, ch_env) <= (int64_t)g0_wgs ? ch_env : 0u;
        if (cdiv((int64_t)cnt, 2048) <= (int64_t)g0_wgs) return 2048u;
        if (cdiv((int64_t)cnt, 4096) <= (int64_t)g0_wgs) return 4096u;
        return 0u;
    };
    // The early / late split (k_sk_classify): a level's early part is keyed and sorted on a second stream while the level
    // below is being flooded; behind that level only the late part is keyed (k_sk_keys) and everything stamped in one launch
    // (k_sk_split_assign).  IVX_SK_SPLIT=0: every level's generation 0 as one list on the chain's own stream (A/B, tests).
    const char *esp = getenv("IVX_SK_SPLIT");
    const char *penv0 = getenv("IVX_SK_PERSIST");
    const bool split_on = !(esp && esp[0] == '0') && !sort_fused && !(penv0 && penv0[0] == '0');
    const char *elm = getenv("IVX_SK_LATE_MAX"); // longest late part ranked by brute force (tests: 0 = every late part through the sort)
    const uint32_t late_max = elm ? std::min<uint32_t>((uint32_t)atoll(elm), LATE_MAX) : LATE_MAX;
    {
        void *mem2 = nullptr;
        const size_t need = sk_layout2(ngen0, maxcnt, split_on, nullptr, &b);
        IVX_REQUIRE(ws_get_s(WS_WSSK, st, need, &mem2) == IVX_OK, IVX_ENOMEM, "watershed: %zu bytes of scratch", need);
        sk_layout2(ngen0, maxcnt, split_on, (char *)mem2, &b);
    }
    // keys -> sorted chunks -> (pairwise ranks | merge passes) of one list on one stream; what comes back is read as
    // "position of entry i = (i & (ch - 1)) + sum over the share planes" (merge passes: no planes, ch covers the whole list)
    struct SortOut {
        const unsigned long long *key;
        const uint32_t *val, *part;
        unsigned shares;
        uint32_t ch;
    };
    auto sort_list = [&](hipStream_t s, SkSortBufs &w, const uint32_t *el, uint32_t cnt, uint32_t climit, SortOut *out) -> int {
        const unsigned gb = (unsigned)cdiv(cnt, 256);
        WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_sk_keys<CC, MT>), dim3(gb), dim3(256), 0, s, g, b.C, mk, I, b.comp, b.tau, el, w.key_a, w.val_a, cnt, climit));
        IVX_LAUNCH_CHECK();
        const uint32_t chb = chunk_len(cnt);
        const int64_t nch = cdiv((int64_t)cnt, chb);
        const bool pairs = !sort_merge && (nch <= PAIR_CHUNKS || ch_env);
        unsigned long long *ka = w.key_a, *kb = w.key_b;
        uint32_t *va = w.val_a, *vb = w.val_b;
        if (chb <= 512) hipLaunchKernelGGL(k_sk_sort_chunks<64>, dim3((unsigned)nch), dim3(64), 0, s, ka, va, kb, vb, cnt, chb);
        else if (chb <= 1024) hipLaunchKernelGGL(k_sk_sort_chunks<128>, dim3((unsigned)nch), dim3(128), 0, s, ka, va, kb, vb, cnt, chb);
        else hipLaunchKernelGGL(k_sk_sort_chunks<256>, dim3((unsigned)nch), dim3(256), 0, s, ka, va, kb, vb, cnt, chb);
        IVX_LAUNCH_CHECK();
        std::swap(ka, kb);
        std::swap(va, vb);
        out->part = w.rank;
        if (pairs) {
            unsigned shares = 0;
            if (nch > 1) { // (chunk, share of the other chunks): a thousand or two workgroups
                shares = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nch - 1, PAIR_WGS / nch), (int64_t)(PLANE_CAP / cnt)));
                if (chb <= 512) hipLaunchKernelGGL(k_sk_rank_pairs<2>, dim3((unsigned)nch, shares), dim3(256), 0, s, ka, w.rank, cnt, chb);
                else if (chb <= 1024) hipLaunchKernelGGL(k_sk_rank_pairs<4>, dim3((unsigned)nch, shares), dim3(256), 0, s, ka, w.rank, cnt, chb);
                else hipLaunchKernelGGL(k_sk_rank_pairs<8>, dim3((unsigned)nch, shares), dim3(256), 0, s, ka, w.rank, cnt, chb);
                IVX_LAUNCH_CHECK();
            }
            out->shares = shares;
            out->ch = chb;
        } else { // too many chunks to rank pairwise: merge passes
            for (uint64_t run = chb; run < cnt; run *= 2) {
                const uint32_t tile = (uint32_t)std::min<uint64_t>(MP_TILE, 2 * run);
                hipLaunchKernelGGL(k_sk_merge_pass, dim3((unsigned)cdiv((int64_t)cnt, tile)), dim3(256), 0, s, ka, va, kb, vb, cnt, (uint32_t)run, tile);
                IVX_LAUNCH_CHECK();
                std::swap(ka, kb);
                std::swap(va, vb);
            }
            out->shares = 0;
            out->ch = 0x80000000u; // (a power of two above every list: position = index)
        }
        out->key = ka;
        out->val = va;
        return IVX_OK;
    };

    tm.mark(st);
    // ---- 3. the level chain ----------------------------------------------------------------------------------
    int64_t nlevels = 0, ntile_rounds = 0, nsmall_runs = 0;
    const bool trace = getenv("IVX_WS_TRACE") != nullptr;
    const char *epb = getenv("IVX_SK_PER_WG"); // list entries per working workgroup (A/B measurements)
    const uint32_t per_wg = epb && atoi(epb) >= 32 ? (uint32_t)atoi(epb) : 256u; // (round 6: 1024 -> 256 = one pass per workgroup, 49.8 -> 44.7 ms of level chain at 512^3)
    const char *eso = getenv("IVX_SK_SOLO"); // most list entries one workgroup takes alone inside a resident launch (0: never; A/B)
    const uint32_t solo_max = eso ? (uint32_t)atoi(eso) : SK_SOLO_MAX;
    const char *elo = getenv("IVX_SK_LOCAL"), *elw = getenv("IVX_SK_LOCAL_WGS"); // the level's rounds on one XCD (k_sk_level<.., LOCAL>); A/B
    const bool level_local = elo && elo[0] == '1';
    const int64_t local_wgs = elw && atoi(elw) >= 1 ? std::min(atoi(elw), 128) : 128; // workgroups wanted on that XCD (32 CUs; 256 -- a launch of 2 048 -- never got all its workgroups started beside the side stream)
    const char *erc = getenv("IVX_SK_RES_PER_CU");
    const int64_t res_per_cu = erc && atoi(erc) >= 1 && atoi(erc) <= 4 ? atoi(erc) : 1;
    uint32_t start = 0, dstart = 0, roff = 0, gbase = 1, seq = 0;
    // IVX_SK_PERSIST=0: a level's rounds as separate launches (k_sk_round, queued in batches; A/B measurements).  Default: one
    // resident launch per level (k_sk_level) and no host read between consecutive levels -- the next level's first
    // generation travels on the device (SkState::gnext); `gknown` says whether the host's gbase is current.
    const char *penv = getenv("IVX_SK_PERSIST");
    const bool persist = !(penv && penv[0] == '0');
    bool gknown = true;
    auto sync_gbase = [&]() -> int { // the chain's last level has finished: fetch the generation counter
        if (gknown) return IVX_OK;
        uint32_t mseq = 0, msg[4] = {0, 0, 0, 0};
        int rc = mailbox_publish(&b.st->done, 4, st, &mseq);
        if (rc != IVX_OK) return rc;
        rc = mailbox_wait(mseq, st, msg, 4);
        if (rc != IVX_OK) return rc;
        IVX_REQUIRE(msg[0] == 1, IVX_EINVAL, "watershed: a level's resident launch lost its hand-over (state %u)", msg[0]);
        gbase = msg[1] + 1;
        gknown = true;
        return IVX_OK;
    };
    SkLists lists;
    for (int i = 0; i < 2; i++) lists.l[i] = b.lists[i];
    const char *senv = getenv("IVX_SK_SMALL"); // 0: never take the one-workgroup path (A/B measurements)
    const bool small_on = !(senv && senv[0] == '0');
    auto is_small = [&](uint32_t c) { return small_on && hist[c] <= (uint32_t)SMALL_GEN0 && lhist[c] <= SMALL_TOTAL; };
    SkSmallArgs sa{nullptr, 0u, b.C, I, b.comp, b.pmask, b.zmask, b.elist, /* dcursor holds positions in the one list: */ b.elist, b.hist, b.cursor, b.dhist, b.dcursor, b.tau, b.runlabel,
                   b.lists[0], b.lists[1], b.st};
    static const char *tenv = getenv("IVX_SK_TILE_LEVEL"); // voxels from which a basin-free level is relaxed tile-wise (A/B; 0 = never)
    const uint64_t tile_min = tenv ? (uint64_t)atoll(tenv) : ((uint64_t)1 << 16);
    auto is_tile_level = [&](uint32_t c) { return dhist[c] == 0 && tile_min && lhist[c] >= tile_min; };
    // ---- the second stream (the early parts' sorts) and the events that order it against the chain
    struct SkSide {
        hipStream_t stream = nullptr;
        hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr}, early[2] = {nullptr, nullptr};
    };
    static thread_local SkSide side;
    if (split_on && !side.stream) {
        IVX_HIP(hipStreamCreateWithFlags(&side.stream, hipStreamNonBlocking));
        for (auto &e : side.done) IVX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : side.early) IVX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    static thread_local bool attr_set[2] = {false, false};
    if (split_on && !attr_set[sizeof(MT) == 2]) { // (one instantiation per marker type)
        IVX_HIP(hipFuncSetAttribute((const void *)k_sk_split_assign<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LATE_MAX * 12)));
        attr_set[sizeof(MT) == 2] = true;
    }
    uint32_t ndone = 0;           // events recorded on the chain's stream so far (a ring of four)
    uint32_t early_of = 0xFFFFFFFFu, early_cnt = 0; // the level whose early part is (being) sorted on the side stream, in set early_set
    int early_set = 0, next_set = 0;
    int64_t nsplit = 0;
    // key records of a level's late part, two sets like the early parts' (6 neighbours; at the end of kind[]'s bytes, which nobody
    // reads after the buckets; IVX_SK_LATE_RECS=0: the late keys chase the neighbours themselves)
    static const bool late_recs_on = []() { const char *e = getenv("IVX_SK_LATE_RECS"); return !(e && e[0] == '0'); }();
    const size_t late_rec_bytes = ((size_t)late_max * 32 + 255) & ~(size_t)255;
    uint32_t *late_recs[2] = {nullptr, nullptr};
    if (late_recs_on && split_on && conn == 6 && (size_t)g.n >= (size_t)g.n / 8 + 1024 + 2 * late_rec_bytes + ((size_t)1 << 22))
        for (int q = 0; q < 2; q++) late_recs[q] = (uint32_t *)(b.kind + (((size_t)g.n - (size_t)(2 - q) * late_rec_bytes) & ~(size_t)255));
    // the early part of the level that follows c, if that level is a candidate: called before c's own work is queued
    auto queue_early_after = [&](uint32_t c) -> int {
        if (!split_on) return IVX_OK;
        uint32_t cn = c + 1;
        while (cn < 65535 && !hist[cn]) cn++;
        if (cn >= 65535) return IVX_OK;
        const uint32_t ce = hist[cn] - hist_l[cn];
        if (is_small(cn) || is_tile_level(cn) || ce == 0 || hist_l[cn] > hist[cn] / 2) return IVX_OK; // (mostly late: nothing to gain)
        uint32_t sn = start + hist[c]; // where level cn's stretch of the list begins (levels between c and cn are empty)
        // levels below c are final -- and c's own generation 0 (the chain's critical stretch) has the chip to itself: the side
        // stream starts when c's flood does
        IVX_HIP(hipEventRecord(side.done[ndone % 4], st));
        ndone++;
        IVX_HIP(hipStreamWaitEvent(side.stream, side.done[(ndone - 1) % 4], 0));
        SortOut so;
        const int rc = sort_list(side.stream, b.sort[1], b.elist + sn, ce, c, &so); // (neighbours below c count: c itself is being flooded)
        if (rc != IVX_OK) return rc;
        hipLaunchKernelGGL(k_sk_scatter_sorted, dim3((unsigned)cdiv(ce, 256)), dim3(256), 0, side.stream, so.key, so.val, so.part, so.shares, so.ch,
                           b.ekey[next_set], b.eval[next_set], ce);
        IVX_LAUNCH_CHECK();
        if (late_recs[next_set] && hist_l[cn] && hist_l[cn] <= late_max) { // the late part's key records (k_sk_small_recs), for the chain's k_sk_keys_rec
            hipLaunchKernelGGL(k_sk_small_recs<MT>, dim3((unsigned)cdiv(hist_l[cn], 256)), dim3(256), 0, side.stream, g, b.C, mk, I, b.comp,
                               b.elist + sn + ce, hist_l[cn], late_recs[next_set]);
            IVX_LAUNCH_CHECK();
        }
        IVX_HIP(hipEventRecord(side.early[next_set], side.stream));
        early_of = cn;
        early_cnt = ce;
        early_set = next_set;
        next_set ^= 1;
        return IVX_OK;
    };
    for (uint32_t c = 0; c < 65535; c++) {
        const uint32_t cnt = hist[c], ndl = dhist[c];
        if (!cnt) { // (no voxel of the level at all: a level's drained voxels hang off its generation 0)
            dstart += ndl;
            continue;
        }
        if (is_small(c)) { // a run of consecutive small levels: one launch
            {
                const int rc = sync_gbase();
                if (rc != IVX_OK) return rc;
            }
            uint32_t c_hi = c, esum = 0, dsum = 0, nlv = 0;
            for (uint32_t q = c; q < 65535 && (hist[q] == 0 || is_small(q)); q++)
                if (hist[q]) c_hi = q;
            for (uint32_t q = c; q <= c_hi; q++) {
                nlv += hist[q] != 0;
                esum += hist[q];
                dsum += dhist[q];
            }
            // key records of the run's entries (6 neighbours; in kind[]'s bytes behind the level plane's: nobody reads those any more)
            static const bool recs_on = []() { const char *e = getenv("IVX_SK_SMALL_RECS"); return !(e && e[0] == '0'); }();
            SkSmallArgs sr = sa;
            const size_t rec_off = ((size_t)g.n / 8 + 511) & ~(size_t)255;
            if (recs_on && conn == 6 && esum && rec_off + (size_t)esum * 32 + 2 * late_rec_bytes + 512 <= (size_t)g.n) {
                uint32_t *rec = (uint32_t *)(b.kind + rec_off);
                hipLaunchKernelGGL(k_sk_small_recs<MT>, dim3((unsigned)cdiv(esum, 256)), dim3(256), 0, st, g, b.C, mk, I, b.comp, b.elist + start, esum, rec);
                IVX_LAUNCH_CHECK();
                sr.rec = rec;
                sr.run_start = start;
            }
            hipLaunchKernelGGL(k_sk_levels_small<MT>, dim3(1), dim3(1024), 0, st, g, sr, mk, c, c_hi, gbase, roff);
            IVX_LAUNCH_CHECK();
            uint32_t mseq = 0, msg[4] = {0, 0, 0, 0};
            int rc = mailbox_publish(&b.st->done, 4, st, &mseq);
            if (rc != IVX_OK) return rc;
            rc = mailbox_wait(mseq, st, msg, 4);
            if (rc != IVX_OK) return rc;
            gbase = msg[1] + 1;
            roff += esum;
            start += esum;
            dstart += dsum;
            nlevels += nlv;
            nsmall_runs++;
            if (trace) fprintf(stderr, "sk levels %u..%u (%u levels) in one workgroup -> generation %u\n", c, c_hi, nlv, gbase - 1);
            c = c_hi;
            continue;
        }
        nlevels++;
        const auto lvl_t0 = std::chrono::steady_clock::now();
        uint32_t lvl_batches = 0;
        const unsigned gb = (unsigned)cdiv(cnt, 256);
        const bool tile_level = is_tile_level(c);
        const bool split_here = early_of == c; // this level's early part is on its way (queued when the level below started)
        const uint32_t my_early_cnt = early_cnt;
        const int my_early_set = early_set;
        if (tile_level || !persist || trace) {
            const int rc = sync_gbase();
            if (rc != IVX_OK) return rc;
        }
        const uint32_t g0_gbase = gknown ? gbase : 0u, g0_seq = persist && !tile_level ? 0u : seq;
        if (const uint32_t chl = g0_chunk(cnt)) { // keys -> sorted -> stamps, run labels, first frontier: one launch
            const unsigned nch = (unsigned)cdiv((int64_t)cnt, chl);
            if (chl <= 2048) {
                WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_sk_gen0<CC, MT, 2>), dim3(nch), dim3(G0_T), 0, st, g, b.C, mk, I, b.comp, b.tau,
                                                          b.elist + start, b.sort[0].key_a, b.runlabel, b.lists[0], cnt, chl, c, roff, g0_gbase, g0_seq,
                                                          b.st, b.g0ctl));
            } else {
                WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_sk_gen0<CC, MT, 4>), dim3(nch), dim3(G0_T), 0, st, g, b.C, mk, I, b.comp, b.tau,
                                                          b.elist + start, b.sort[0].key_a, b.runlabel, b.lists[0], cnt, chl, c, roff, g0_gbase, g0_seq,
                                                          b.st, b.g0ctl));
            }
            IVX_LAUNCH_CHECK();
        } else if (split_here) { // the early part is sorted already: late keys, then every stamp in one launch
            const uint32_t ce = my_early_cnt, cl = cnt - ce;
            nsplit++;
            if (cl <= late_max) { // late keys, then every stamp in one launch (the late pairs ranked by brute force in LDS)
                IVX_HIP(hipStreamWaitEvent(st, side.early[my_early_set], 0)); // (long since: the side stream worked beside the level below)
                if (cl && late_recs[my_early_set]) {
                    hipLaunchKernelGGL(k_sk_keys_rec, dim3((unsigned)cdiv(cl, 256)), dim3(256), 0, st, late_recs[my_early_set], b.tau, b.sort[0].key_a,
                                       b.sort[0].val_a, cl);
                    IVX_LAUNCH_CHECK();
                } else if (cl) {
                    WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_sk_keys<CC, MT>), dim3((unsigned)cdiv(cl, 256)), dim3(256), 0, st, g, b.C, mk, I, b.comp,
                                                              b.tau, b.elist + start + ce, b.sort[0].key_a, b.sort[0].val_a, cl, c));
                    IVX_LAUNCH_CHECK();
                }
                const unsigned nwg = (unsigned)(cdiv(ce, 256) + cdiv(cl, LATE_PER_WG));
                hipLaunchKernelGGL(k_sk_split_assign<MT>, dim3(nwg), dim3(256), (size_t)cl * 12, st, b.ekey[my_early_set], b.eval[my_early_set], ce,
                                   b.sort[0].key_a, b.sort[0].val_a, cl, mk, b.tau, b.runlabel, b.lists[0], roff, g0_gbase, g0_seq, b.st);
                IVX_LAUNCH_CHECK();
            } else { // a late part too long for that (1024^3: 16 000 of a level's 360 000): sorted like any list, placed behind the early part
                SortOut so;
                const int rc = sort_list(st, b.sort[0], b.elist + start + ce, cl, c, &so);
                if (rc != IVX_OK) return rc;
                IVX_HIP(hipStreamWaitEvent(st, side.early[my_early_set], 0));
                hipLaunchKernelGGL(k_sk_assign_ranked<MT>, dim3((unsigned)cdiv(ce, 256)), dim3(256), 0, st, b.ekey[my_early_set], b.eval[my_early_set],
                                   (const uint32_t *)nullptr, 0u, 0x80000000u, mk, b.tau, b.runlabel, b.lists[0], ce, 0u, cnt, roff, g0_gbase, g0_seq, b.st);
                IVX_LAUNCH_CHECK();
                hipLaunchKernelGGL(k_sk_assign_ranked<MT>, dim3((unsigned)cdiv(cl, 256)), dim3(256), 0, st, so.key, so.val, so.part, so.shares, so.ch, mk,
                                   b.tau, b.runlabel, b.lists[0], cl, ce, cnt, roff, g0_gbase, g0_seq, b.st);
                IVX_LAUNCH_CHECK();
            }
            if ((mbits[c >> 5] >> (c & 31u)) & 1u) {
                hipLaunchKernelGGL(k_sk_mixed<MT>, dim3(gb), dim3(256), 0, st, b.lists[0], mk, cnt, b.st);
                IVX_LAUNCH_CHECK();
            }
        } else {
            SortOut so;
            const int rc = sort_list(st, b.sort[0], b.elist + start, cnt, c, &so);
            if (rc != IVX_OK) return rc;
            hipLaunchKernelGGL(k_sk_assign_ranked<MT>, dim3(gb), dim3(256), 0, st, so.key, so.val, so.part, so.shares, so.ch, mk, b.tau, b.runlabel,
                               b.lists[0], cnt, 0u, cnt, roff, g0_gbase, g0_seq, b.st);
            IVX_LAUNCH_CHECK();
            if ((mbits[c >> 5] >> (c & 31u)) & 1u) {
                hipLaunchKernelGGL(k_sk_mixed<MT>, dim3(gb), dim3(256), 0, st, b.lists[0], mk, cnt, b.st);
                IVX_LAUNCH_CHECK();
            }
        }
        {
            const int rc = queue_early_after(c); // (this level's stamps are queued: the next level's early part may start beside its flood)
            if (rc != IVX_OK) return rc;
        }
        if (tile_level) {
            hipLaunchKernelGGL(k_sk_mark_tiles, dim3(gb), dim3(256), 0, st, g, b.lists[0], cnt, b.dirty);
            IVX_LAUNCH_CHECK();
            // the level's voxels as a bit plane (in kind[]'s bytes: nobody reads those after the buckets); IVX_SK_PLANE=0: C and I per cell
            static const bool plane_on = []() { const char *e = getenv("IVX_SK_PLANE"); return !(e && e[0] == '0'); }();
            const unsigned long long *P = nullptr;
            if (plane_on && (((uintptr_t)I | (uintptr_t)b.C) & 15) == 0) {
                hipLaunchKernelGGL(k_sk_level_plane, dim3((unsigned)cdiv(cdiv(g.n, 64) * 8, 256)), dim3(256), 0, st, b.C, I, g.n, c, (unsigned long long *)b.kind);
                IVX_LAUNCH_CHECK();
                P = (const unsigned long long *)b.kind;
            }
            // Rounds of dirty tiles.  The host never stands between two rounds' kernels: a round's relaxation is launched with a grid
            // guessed from the previous round's list (1.5 x + 64: the wave front grows slowly) BEFORE the host has read how long this
            // round's list is -- the read then overlaps the kernel (110 us), and a list longer than the guess gets a second launch
            // behind it.  (Rounds 1 - 5: list length read first, 20 us of host round trip per round, 61 / 125 rounds per flood.)
            uint32_t guess = 0;
            int parity = 0; // (both list counters are zero between two loops: a loop ends on an empty list, which cleared the other one)
            for (;;) {
                uint32_t mseq = 0, nl = 0, *cur = nullptr;
                int rc = ws_build_list_publish(g.ntiles, b.dirty, b.tlist, b.wst, parity, st, &mseq, &cur); // (list, count and mailbox in one launch)
                if (rc != IVX_OK) return rc;
                parity ^= 1;
                if (guess) {
                    WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_sk_plateau_relax<CC>, dim3(guess), dim3(256), 0, st, g, I, b.C, b.tau, b.tlist, b.dirty, c, b.st,
                                                              cur, 0u, P));
                    IVX_LAUNCH_CHECK();
                }
                rc = mailbox_wait(mseq, st, &nl, 1);
                if (rc != IVX_OK) return rc;
                if (!nl) break; // (a guessed launch of this round found an empty list and returned)
                ntile_rounds++;
                if (nl > guess) {
                    WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_sk_plateau_relax<CC>, dim3(nl - guess), dim3(256), 0, st, g, I, b.C, b.tau, b.tlist, b.dirty, c,
                                                              b.st, cur, guess, P));
                    IVX_LAUNCH_CHECK();
                }
                guess = (uint32_t)std::min<int64_t>(g.ntiles, (int64_t)nl + nl / 2 + 64);
            }
            uint32_t mseq = 0, msg[4] = {0, 0, 0, 0};
            int rc = mailbox_publish(&b.st->done, 4, st, &mseq);
            if (rc != IVX_OK) return rc;
            rc = mailbox_wait(mseq, st, msg, 4);
            if (rc != IVX_OK) return rc;
            gbase = msg[1] + 1;
            roff += cnt;
            start += cnt;
            dstart += ndl;
            if (trace)
                fprintf(stderr, "sk level %u gen0 %u tile-wise -> generation %u, %.0f us\n", c, cnt, gbase - 1,
                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - lvl_t0).count());
            continue;
        }
        if (persist) { // one resident launch; the host moves on without reading anything
            // (as many workgroups as twice the first frontier asks for, at most res_per_cu per compute unit: all of them must be
            // resident -- 256 threads, 30 registers and 16 KB of LDS each leave room for eight)
            // (a large first frontier -- the levels of a 1024^3 volume hold 4 - 5 x 10^5 generation-0 voxels -- is served better by
            // twice the workgroups with half the entries each: 235 -> 224 ms at 1024^3; at 512^3, 5 x 10^4 voxels, it measured worse)
            const bool wide = !epb && !erc && cnt >= (1u << 17);
            const uint32_t lvl_per_wg = per_wg; // (round 6: one pass per workgroup here too -- 512 entries measured 136.7 ms of level chain at 1024^3, 256: 131.6)
            const int64_t lvl_res = wide ? 2 : res_per_cu;
            const unsigned nres = (unsigned)std::min<int64_t>(std::max<int64_t>(cdiv(2 * (int64_t)std::max(cnt, ndl), lvl_per_wg), 8), lvl_res * std::max(ncu, 8));
            if (level_local) { // the rounds on one XCD: an eighth of an eight times wider launch
                const unsigned want_wgs = (unsigned)std::min<int64_t>(std::max<int64_t>(cdiv(2 * (int64_t)std::max(cnt, ndl), lvl_per_wg), 1), local_wgs);
                WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_sk_level<CC, true>), dim3(8 * want_wgs), dim3(256), 0, st, g, b.pmask, b.zmask, b.comp, b.tau,
                                                          lists, b.dlist + dstart, droot_d + dstart, ndl, lvl_per_wg, solo_max, b.st));
            } else {
                WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_sk_level<CC, false>), dim3(nres), dim3(256), 0, st, g, b.pmask, b.zmask, b.comp, b.tau, lists,
                                                          b.dlist + dstart, droot_d + dstart, ndl, lvl_per_wg, solo_max, b.st));
            }
            IVX_LAUNCH_CHECK();
            gknown = false;
            if (trace) {
                const int rc = sync_gbase();
                if (rc != IVX_OK) return rc;
                fprintf(stderr, "sk level %u gen0 %u drained %u -> generation %u, resident, %.0f us\n", c, cnt, ndl, gbase - 1,
                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - lvl_t0).count());
            }
            roff += cnt;
            start += cnt;
            dstart += ndl;
            continue;
        }
        // rounds are queued in growing batches; one host read per batch (a round after the level's last returns at once)
        // (a host read costs about as much as eight idle launches: the first batch is sized for a typical level)
        uint32_t batch = 16, width = cnt;
        for (;;) {
            // the frontier can grow a lot inside one batch: the grid is sized for a large one (idle workgroups leave at once)
            const unsigned nb = (unsigned)std::min<int64_t>(std::max<int64_t>(4 * cdiv(std::max(width, ndl), 256), 1024), 4096);
            for (uint32_t r = 0; r < batch; r++) {
                hipLaunchKernelGGL(k_sk_round, dim3(nb), dim3(256), 0, st, g, b.pmask, b.zmask, b.comp, b.tau, lists, b.dlist + dstart, ndl, seq++, per_wg, b.st);
                IVX_LAUNCH_CHECK();
            }
            uint32_t mseq = 0, msg[4] = {0, 0, 0, 0};
            int rc = mailbox_publish(&b.st->done, 4, st, &mseq);
            if (rc != IVX_OK) return rc;
            rc = mailbox_wait(mseq, st, msg, 4);
            if (rc != IVX_OK) return rc;
            gbase = msg[1];
            lvl_batches++;
            if (msg[0]) break;
            width = std::max(msg[3], 1u);
            batch = std::min(batch * 2, 64u);
        }
        if (trace)
            fprintf(stderr, "sk level %u gen0 %u drained %u -> generation %u, %u batches, %.0f us\n", c, cnt, ndl, gbase, lvl_batches,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - lvl_t0).count());
        gbase++;
        IVX_REQUIRE(gbase < 0x7FFFFFF0u, IVX_EINVAL, "watershed: more than 2^31 generations");
        roff += cnt;
        start += cnt;
        dstart += ndl;
    }

    {
        const int rc = sync_gbase();
        if (rc != IVX_OK) return rc;
        IVX_REQUIRE(gbase < 0x7FFFFFF0u, IVX_EINVAL, "watershed: more than 2^31 generations");
    }
    tm.mark(st);
    // ---- 4. labels -------------------------------------------------------------------------------------------
    hipLaunchKernelGGL(k_sk_labels<MT>, dim3(gl), dim3(256), 0, st, g.n, b.comp, b.tau, b.runlabel, out, out32, out8);
    IVX_LAUNCH_CHECK();
    tm.mark(st);
    SkState hs;
    SkG0Ctl hg;
    IVX_HIP(hipMemcpyAsync(&hs, b.st, sizeof(hs), hipMemcpyDeviceToHost, st));
    IVX_HIP(hipMemcpyAsync(&hg, b.g0ctl, sizeof(hg), hipMemcpyDeviceToHost, st));
    IVX_HIP(hipStreamSynchronize(st));
    IVX_REQUIRE(!hg.fail, IVX_EHIP, "watershed: a generation-0 launch lost its device-wide barrier (workgroups not resident)");
    if (trace) fprintf(stderr, "sk levels whose early part was sorted beside the level below: %lld of %lld\n", (long long)nsplit, (long long)nlevels);
    if (trace && SK_TICKS)
        fprintf(stderr, "sk rounds, workgroup 0 (%u rounds): A: loads %.0f us, plateau offers %.0f, basin offers %.0f, flush %.0f, drain %.0f | B: loads %.0f, offers %.0f, flush %.0f, drain %.0f | word wait %.0f, ticket %.0f\n",
                hs.ticks[14], hs.ticks[0] * 0.01, hs.ticks[1] * 0.01, hs.ticks[2] * 0.01, hs.ticks[3] * 0.01, hs.ticks[4] * 0.01, hs.ticks[6] * 0.01,
                hs.ticks[7] * 0.01, hs.ticks[9] * 0.01, hs.ticks[10] * 0.01, hs.ticks[12] * 0.01, hs.ticks[13] * 0.01);
    if (trace)
        fprintf(stderr, "sk generation-0 launches, workgroup 0: keys %.0f us, chunk sort %.0f, barrier %.0f, ranks %.0f, stamps %.0f (sums over the flood)\n",
                hg.ticks[0] * 0.01, hg.ticks[1] * 0.01, hg.ticks[2] * 0.01, hg.ticks[3] * 0.01, hg.ticks[4] * 0.01);
    if (stats) {
        stats[0] = rounds; stats[1] = visits; stats[2] = nlevels; stats[3] = gbase; stats[4] = M; stats[5] = (int64_t)ngen0;
        stats[6] = hs.mixed; stats[7] = hs.rounds;
        for (int i = 8; i < 16; i++) stats[i] = 0;
        tm.read(stats + 8); // [8] costs, [9] generation 0, [10] level chain, [11] labels (microseconds)
        stats[12] = hs.brounds; stats[13] = hs.gens; stats[14] = nsmall_runs; stats[15] = ntile_rounds;
    }
    return IVX_OK;
}

} // namespace

extern "C" int ivx_dev_watershed_sk(const uint16_t *image, int mdtype, const void *markers, int64_t dz, int64_t dy, int64_t dx,
                                    const uint8_t strct[27], void *out_labels, int32_t *out_i32, uint8_t *out_u8, uint16_t *cost_out,
                                    int64_t stats[16], void *stream) {
    WsGeom g;
    const int rc = make_geom(dz, dy, dx, strct, &g);
    if (rc != IVX_OK) return rc;
    IVX_REQUIRE(mdtype == IVX_I16 || mdtype == IVX_I8, IVX_EINVAL, "watershed: markers must be int16 or int8");
    IVX_REQUIRE(image && markers && (out_labels || out_i32 || out_u8), IVX_EINVAL, "watershed: null buffer");
    if (mdtype == IVX_I16)
        return sk_run<int16_t>(g, image, (const int16_t *)markers, (int16_t *)out_labels, out_i32, out_u8, cost_out, stats, S(stream));
    return sk_run<int8_t>(g, image, (const int8_t *)markers, (int8_t *)out_labels, out_i32, out_u8, cost_out, stats, S(stream));
}

__global__ void k_sk_widen(const uint8_t *__restrict__ in, uint16_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// host arrays in, int32 labels out (scikit-image's output dtype)
extern "C" int ivx_watershed_sk(int idtype, const void *input, const int64_t shape[3], int mdtype, const void *markers,
                                const uint8_t strct[27], int32_t *output, uint16_t *cost_out, int64_t stats[16]) {
    HostCallGuard guard;
    IVX_REQUIRE(idtype == IVX_U8 || idtype == IVX_U16, IVX_EINVAL, "watershed: image must be uint8 or uint16");
    IVX_REQUIRE(mdtype == IVX_I16 || mdtype == IVX_I8, IVX_EINVAL, "watershed: markers must be int16 or int8");
    const int64_t n = shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    const size_t msz = mdtype == IVX_I16 ? 2 : 1;
    void *dI = nullptr, *dM = nullptr, *dO = nullptr, *dC = nullptr, *dT = nullptr;
    int rc;
    if ((rc = ws_get(WS_IN, (size_t)n * 2, &dI)) != IVX_OK) return rc;
    if ((rc = ws_get(WS_AUX0, (size_t)n * msz, &dM)) != IVX_OK) return rc;
    if ((rc = ws_get(WS_OUT, (size_t)n * 4, &dO)) != IVX_OK) return rc;
    if (cost_out && (rc = ws_get(WS_AUX1, (size_t)n * 2, &dC)) != IVX_OK) return rc;
    if (idtype == IVX_U8) {
        if ((rc = ws_get(WS_AUX2, (size_t)n, &dT)) != IVX_OK) return rc;
        IVX_HIP(hipMemcpy(dT, input, (size_t)n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_sk_widen, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, 0, (const uint8_t *)dT, (uint16_t *)dI, n);
        IVX_LAUNCH_CHECK();
    } else {
        IVX_HIP(hipMemcpy(dI, input, (size_t)n * 2, hipMemcpyHostToDevice));
    }
    IVX_HIP(hipMemcpy(dM, markers, (size_t)n * msz, hipMemcpyHostToDevice));
    rc = ivx_dev_watershed_sk((const uint16_t *)dI, mdtype, dM, shape[0], shape[1], shape[2], strct, nullptr, (int32_t *)dO, nullptr,
                              (uint16_t *)dC, stats, nullptr);
    if (rc != IVX_OK) return rc;
    IVX_HIP(hipMemcpy(output, dO, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (cost_out) IVX_HIP(hipMemcpy(cost_out, dC, (size_t)n * 2, hipMemcpyDeviceToHost));
    return IVX_OK;
}
