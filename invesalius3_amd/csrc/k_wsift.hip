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
#include "ws_tiles.h"

namespace {
template <typename MT> struct WsEntryPred { // entries that are not markers (those are level 0, listed apart)
    const uint32_t *comp;
    const MT *mk;
    __device__ bool operator()(int64_t p) const { return comp[p] == ENTRY && mk[p] == 0; }
};

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
    load_tile<false>(g, z0, y0, x0, I, C, s);
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

__device__ __forceinline__ uint32_t ws_tau_of(const uint32_t *__restrict__ comp, const uint32_t *tau, int64_t v) {
    const uint32_t cv = comp[v];
    return __hip_atomic_load(&tau[cv == ENTRY ? (uint32_t)v : cv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// mark the wave's keys in the used-key bitmap: one atomic per distinct key of the wave, and none for keys already marked (the
// leader looks right before it would set the bit).  (Round 5, measured: every lane testing its own bit up front -- all loads in
// flight together -- and fire-and-forget atomics for the unmarked keys made the level chain SLOWER, 11.3 -> 13.2 ms at 512^3:
// when a level starts, a thousand workgroups see the same few hundred bits unset at once and all of them send their atomics;
// the late look inside the loop is what keeps the same-address traffic down.)
__device__ __forceinline__ void ws_mark_used(uint32_t *used, bool has, uint32_t K) {
    unsigned long long todo = __ballot(has);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lk = __shfl(K, leader, 64);
        const unsigned long long same = __ballot(has && K == lk);
        if (lane == leader) {
            const uint32_t bit = 1u << (lk & 31);
            if (!(__hip_atomic_load(&used[lk >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&used[lk >> 5], bit);
        }
        todo &= ~same;
    }
}

// ... per WORKGROUP when the stamps issued so far fit an LDS bitmap (2^18 of them; a 512^3 flood issues ~10^5): the lanes set
// their keys' bits with LDS atomics, then every lane flushes a few words -- its loads of the global words in flight together,
// an atomic only where bits are still missing.  The per-wave loop above pays one dependent agent-scope load per DISTINCT key
// of the wave, serially (rocprofv3, round 5: k_ws_keys 55 us per level with three gathers per entry, 61 us with one -- the
// gathers were never the cost).
constexpr uint32_t USED_LDS_WORDS = 8192;
__device__ __forceinline__ void ws_mark_used_block(uint32_t *used, uint32_t *s_used, uint32_t nw, bool has, uint32_t K) {
    if (has) atomicOr(&s_used[K >> 5], 1u << (K & 31));
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nw; i += 256) {
        const uint32_t m = s_used[i];
        if (m && (m & ~__hip_atomic_load(&used[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicOr(&used[i], m);
    }
}

// key of every entry of level c: the earliest time stamp among its admissible parents (the set bits of pmask); mark the
// key as used
template <int CONN>
__global__ __launch_bounds__(256) void k_ws_keys(WsGeom g, const uint32_t *__restrict__ pmask, const uint32_t *__restrict__ comp,
                                                 const uint32_t *tau, const uint32_t *__restrict__ elist, uint32_t *__restrict__ key,
                                                 uint32_t *used, uint32_t start, uint32_t count, const WsState *st) {
    __shared__ uint32_t s_used[USED_LDS_WORDS];
    const uint32_t nw = (st->base + 31) >> 5; // (stamps issued before this level: every key is one of them)
    const bool lds = nw <= USED_LDS_WORDS;
    if (lds) {
        for (uint32_t w = threadIdx.x; w < nw; w += 256) s_used[w] = 0u;
        __syncthreads();
    }
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
    if (lds) ws_mark_used_block(used, s_used, nw, act && K != NONE, K);
    else ws_mark_used(used, act && K != NONE, K);
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

// ---- links resolved once, in raster order (round 5) --------------------------------------------------------------------------
// The per-level kernels above chase, for every entry of a level, pmask[p] -> comp[q] -> tau[q or root] (keys) and zmask[p] ->
// comp[q] -> tau[root] (claim): three dependent gathers into 0.5 GB tables at places that have nothing to do with each other
// inside a level's list (PMC: 9.7 + 7.3 GB per 512^3 flood, ~300 B per entry).  But WHICH words of tau an entry reads and
// writes is fixed once the zones are flattened: the scatter pass that puts the entries into their levels' lists walks the
// volume in raster order, where p's masks are a streaming read and comp[p +- 1 / w / hw] sit on lines its neighbours in the
// wave just touched, and leaves next to elist[slot]
// one 32-byte record per entry in place of elist[slot]: the voxel, the (up to two distinct) words of tau that hold its parents'
// stamps, and the (up to three distinct) zone roots it hands its own stamp to; an entry with more of either gets LINK_MORE in
// the last slot and takes the old walk over its mask.  A level's keys / claim then cost one coalesced read of the records and
// ONE gather / atomic each into tau.  (Separate link arrays -- six scattered 4 / 8-byte stores per entry -- cost the scatter
// +1.6 ms at 512^3, +11 ms at 1024^3; one record is two stores into one 32-byte sector.)
constexpr uint32_t LINK_MORE = 0xFFFFFFFEu;

template <int CONN, typename MT>
__global__ __launch_bounds__(256) void k_ws_bucket_links(WsGeom g, const uint16_t *__restrict__ C, const uint32_t *__restrict__ comp,
                                                         const MT *__restrict__ mk, const uint32_t *__restrict__ pmask,
                                                         const uint32_t *__restrict__ zmask, uint32_t *__restrict__ cursor,
                                                         uint32_t *__restrict__ rec, int lb) {
    extern __shared__ uint32_t sh[]; // lb counters (k_ws_bucket: the host has the levels' histogram by now and asks for those that exist)
    for (int i = threadIdx.x; i < lb; i += 256) sh[i] = 0;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * (256 * BK_CH);
    for (int pass = 0; pass < 2; pass++) { // (the counting / placing scheme of k_ws_bucket<PRED, true>)
        for (int j = 0; j < BK_CH; j++) {
            const int64_t p = b0 + (int64_t)j * 256 + threadIdx.x;
            const bool e = p < g.n && comp[p] == ENTRY && mk[p] == 0;
            if (!e) continue;
            const uint32_t c = C[p];
            uint32_t off;
            if (c < (uint32_t)lb) {
                off = atomicAdd(&sh[c], 1u);
                if (pass == 0) continue;
            } else {
                if (pass == 0) continue;
                off = atomicAdd(&cursor[c], 1u);
            }
            uint32_t pl[2] = {NONE, NONE}, zl[3] = {NONE, NONE, NONE};
            int np = 0, nzl = 0;
            uint32_t pm = pmask[p];
            while (pm) {
                const int k = __ffs(pm) - 1;
                pm &= pm - 1;
                const int64_t q = p + (k / 9 - 1) * g.hw + ((k / 3) % 3 - 1) * g.w + (k % 3 - 1);
                const uint32_t cv = comp[q];
                const uint32_t loc = cv == ENTRY ? (uint32_t)q : cv;
                if (loc == pl[0] || loc == pl[1]) continue;
                if (np < 2) pl[np] = loc;
                np++;
            }
            if (np > 2) pl[1] = LINK_MORE;
            uint32_t zm = zmask[p];
            while (zm) {
                const int k = __ffs(zm) - 1;
                zm &= zm - 1;
                const int64_t q = p + (k / 9 - 1) * g.hw + ((k / 3) % 3 - 1) * g.w + (k % 3 - 1);
                const uint32_t r = comp[q];
                if (r == ENTRY || r == zl[0] || r == zl[1] || r == zl[2]) continue;
                if (nzl < 3) zl[nzl] = r;
                nzl++;
            }
            if (nzl > 3) zl[2] = LINK_MORE;
            // ONE 32-byte record per entry (two 16-byte stores into one sector): { p, parent 0, parent 1, zone 0 | zone 1, zone 2, -, - }
            uint4 *r4 = reinterpret_cast<uint4 *>(rec + 8 * (size_t)off);
            r4[0] = make_uint4((uint32_t)p, pl[0], pl[1], zl[0]);
            r4[1] = make_uint4(zl[1], zl[2], 0u, 0u);
        }
        __syncthreads();
        if (pass == 0) {
            for (int i = threadIdx.x; i < lb; i += 256) {
                const uint32_t v = sh[i];
                if (v) sh[i] = atomicAdd(&cursor[i], v);
            }
            __syncthreads();
        }
    }
}

template <int CONN>
__global__ __launch_bounds__(256) void k_ws_keys_links(WsGeom g, const uint32_t *__restrict__ pmask, const uint32_t *__restrict__ comp,
                                                       const uint32_t *tau, const uint32_t *__restrict__ rec, uint32_t *__restrict__ key,
                                                       uint32_t *used, uint32_t start, uint32_t count, const WsState *st) {
    __shared__ uint32_t s_used[USED_LDS_WORDS];
    const uint32_t nw = (st->base + 31) >> 5; // (stamps issued before this level: every key is one of them)
    const bool lds = nw <= USED_LDS_WORDS;
    if (lds) {
        for (uint32_t w = threadIdx.x; w < nw; w += 256) s_used[w] = 0u;
        __syncthreads();
    }
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < count;
    uint32_t K = NONE;
    if (act) {
        const uint4 r0 = *reinterpret_cast<const uint4 *>(rec + 8 * (size_t)(start + i));
        const uint2 l = make_uint2(r0.y, r0.z);
        if (l.y == LINK_MORE) { // more than two distinct parents: the walk over the mask
            const int64_t p = r0.x;
            uint32_t pm = pmask[p];
            while (pm) {
                const int k = __ffs(pm) - 1;
                pm &= pm - 1;
                const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
                K = min(K, ws_tau_of(comp, tau, p + dz * g.hw + dy * g.w + dx));
            }
        } else { // both gathers in flight together
            const uint32_t t0 = l.x != NONE ? __hip_atomic_load(&tau[l.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : NONE;
            const uint32_t t1 = l.y != NONE ? __hip_atomic_load(&tau[l.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : NONE;
            K = min(t0, t1);
        }
        key[start + i] = K;
    }
    if (lds) ws_mark_used_block(used, s_used, nw, act && K != NONE, K);
    else ws_mark_used(used, act && K != NONE, K);
}

template <int CONN>
__global__ __launch_bounds__(256) void k_ws_claim_links(WsGeom g, const uint32_t *__restrict__ zmask, const uint32_t *__restrict__ comp,
                                                        uint32_t *tau, const uint32_t *__restrict__ rec, const uint32_t *__restrict__ key,
                                                        const uint32_t *__restrict__ remap, uint32_t start, uint32_t count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t K = key[start + i];
    if (K == NONE) return;
    const uint4 r0 = *reinterpret_cast<const uint4 *>(rec + 8 * (size_t)(start + i));
    const uint2 r1 = *reinterpret_cast<const uint2 *>(rec + 8 * (size_t)(start + i) + 4);
    const int64_t p = r0.x;
    const uint32_t z0 = r0.w, z1 = r1.x, z2 = r1.y;
    const uint32_t t = remap[K];
    tau[p] = t;
    if (z2 == LINK_MORE) { // more than three distinct zones: the walk over the mask
        uint32_t zm = zmask[p];
        while (zm) {
            const int k = __ffs(zm) - 1;
            zm &= zm - 1;
            const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
            const uint32_t r = comp[p + dz * g.hw + dy * g.w + dx];
            if (r == ENTRY) continue;
            if (__hip_atomic_load(&tau[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > t) atomicMin(&tau[r], t);
        }
        return;
    }
    const uint32_t zz[3] = {z0, z1, z2};
    uint32_t cur[3];
#pragma unroll
    for (int q = 0; q < 3; q++) cur[q] = zz[q] != NONE ? __hip_atomic_load(&tau[zz[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int q = 0; q < 3; q++)
        if (zz[q] != NONE && cur[q] > t) atomicMin(&tau[zz[q]], t);
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

__global__ void k_ws_set_hist0(uint32_t *hist, uint32_t m) { hist[0] = m; }
__global__ void k_ws_set_base(WsState *st, uint32_t m) { st->base = m; }

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
    hipLaunchKernelGGL((k_ws_init<MT, false>), dim3((unsigned)nblk), dim3(256), 0, st, g, mk, I, b.C, b.dirty, b.bcount, b.st);
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
        hipLaunchKernelGGL((k_ws_init<MT, false>), dim3((unsigned)nblk), dim3(256), 0, st, g, mk, I, b.C, b.dirty, b.bcount, b.st);
        IVX_LAUNCH_CHECK();
        const int rc = scan_u32_exclusive(b.bcount, nblk, b.bsum, b.total, st);
        if (rc != IVX_OK) return rc;
    }
    IVX_HIP(hipMemsetAsync(b.used, 0, ((size_t)cap / 32 + 2) * 4, st));

    WsTimer tm;
    tm.on = stats != nullptr;
    tm.mark(st);
    // ---- 1. costs ------------------------------------------------------------------------------------------
    // The levels where the bulk of the volume connects go first, as floods on bit planes (ivx_dev_ws_cost_levels: exact
    // costs for every voxel they reach); the relaxation then starts from final costs around the pockets that are left and
    // has nothing to correct.  IVX_WS_LEVELS=0: relaxation from the markers alone (same costs).
    int64_t rounds = 0, visits = 0;
    int levels_done = 0;
    int64_t level_rounds = 0, level_voxels = 0;
    {
        // (read per call: tests and A/B runs change them)  IVX_WS_LEVELS = most levels (0: off), IVX_WS_LEVELS_FRAC = stop
        // once this share of the voxels is in, IVX_WS_LEVELS_MIN = smallest volume (voxels) that takes this path
        const char *e1 = getenv("IVX_WS_LEVELS"), *e2 = getenv("IVX_WS_LEVELS_FRAC"), *e3 = getenv("IVX_WS_LEVELS_MIN");
        const int lv_max = e1 ? atoi(e1) : 48;
        // (0.6 .. 0.8 measure the same at 512^3; at 1024^3 0.5 / 0.7 / 0.8 / 0.9 give 215 / 223 / 231 / 255 ms: the later levels'
        // floods cross a volume whose planes no longer fit the caches, and cost more than the relaxation they save)
        const double lv_frac = e2 ? atof(e2) : (g.n >= ((int64_t)1 << 29) ? 0.5 : 0.7);
        const int64_t lv_min = e3 ? atoll(e3) : ((int64_t)1 << 21);
        if (lv_max > 0 && conn == 6 && g.w % 64 == 0 && g.h % 16 == 0 && g.n >= lv_min) {
            const int rc = ivx_dev_ws_cost_levels(I, sizeof(MT) == 2 ? IVX_I16 : IVX_I8, mk, g.d, g.h, g.w, b.C, lv_max, lv_frac,
                                                  &levels_done, &level_voxels, &level_rounds, st);
            if (rc != IVX_OK) return rc;
            // the pockets, and anything the levels left: the tiles that hold a voxel without a cost (the levels' costs are final)
            IVX_HIP(hipMemsetAsync(b.dirty, 0, (size_t)g.ntiles, st));
            hipLaunchKernelGGL(k_ws_mark_open_tiles, dim3((unsigned)cdiv(g.n / 8, 256)), dim3(256), 0, st, g, b.C, b.dirty);
            IVX_LAUNCH_CHECK();
        }
        const int rc = ws_cost_rounds<false>(g, conn, I, b.C, b.list, b.dirty, b.pending, b.st, st, &rounds, &visits);
        if (rc != IVX_OK) return rc;
    }
    if (cost_out) IVX_HIP(hipMemcpyAsync(cost_out, b.C, (size_t)g.n * 2, hipMemcpyDeviceToDevice, st));

    tm.mark(st);
    // ---- 2. entries and zones ------------------------------------------------------------------------------
    WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_ws_entries<CC, MT>), dim3((unsigned)g.ntiles), dim3(256), 0, st, g, I, b.C, mk, b.comp, b.pmask, b.zmask));
    IVX_LAUNCH_CHECK();
    {
        const int rc = ws_zone_union(g, conn, b.zmask, b.comp, st);
        if (rc != IVX_OK) return rc;
    }

    tm.mark(st);
    // ---- 3. entries by level -------------------------------------------------------------------------------
    hipLaunchKernelGGL((k_ws_bucket<WsEntryPred<MT>, false>), dim3((unsigned)cdiv(g.n, 256 * BK_CH)), dim3(256), (size_t)BK_LB * 4, st, g.n, b.C, WsEntryPred<MT>{b.comp, mk}, b.hist, b.elist, BK_LB);
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
    hipLaunchKernelGGL(k_ws_fill32, dim3(2048), dim3(256), 0, st, b.tau, g.n, NONE);
    IVX_LAUNCH_CHECK();
    IVX_HIP(hipStreamSynchronize(st)); // hist is on the host now
    // the entries' links next to the list (IVX_WS_LINKS=0: the per-level kernels chase the masks themselves; same labels)
    uint32_t *rec = nullptr;
    {
        const char *el = getenv("IVX_WS_LINKS");
        if (!(el && el[0] == '0')) {
            // one 32-byte record per entry that is not a marker (level 0 -- slots [0, M) of the list -- is never read through
            // them: `rec` points M records in front of the block).  The records are an accelerator, 8x the bytes of the list they
            // replace: when the device cannot spare them the flood takes the unlinked kernels (same labels) instead of failing
            // (ADVICE r5).
            uint64_t total_entries = 0;
            for (uint32_t c = 0; c < 65536; c++) total_entries += hist[c];
            const uint64_t nrec = total_entries > M ? total_entries - M : 0;
            void *lm = nullptr;
            if (ws_get_s(WS_WSLINK, st, (size_t)nrec * 32 + 256, &lm) == IVX_OK && lm) {
                rec = (uint32_t *)((uintptr_t)lm - (uintptr_t)M * 32u);
            } else {
                (void)hipGetLastError(); // (the failed allocation's error is dealt with here)
                if (trace_enabled()) fprintf(stderr, "ivx: watershed_ift: no room for %zu bytes of link records, flooding without them\n", (size_t)nrec * 32);
            }
        }
    }
    int bk_lb = 64; // LDS counters of the placing pass: the levels that have entries (whole 64s, BK_LB at most)
    for (uint32_t c = 0; c < 65536; c++)
        if (hist[c]) bk_lb = (int)std::min<uint32_t>((c + 64u) & ~63u, (uint32_t)BK_LB);
    if (rec) {
        WS_CONN_SWITCH(conn, hipLaunchKernelGGL((k_ws_bucket_links<CC, MT>), dim3((unsigned)cdiv(g.n, 256 * BK_CH)), dim3(256), (size_t)bk_lb * 4, st, g, b.C,
                                                  b.comp, mk, b.pmask, b.zmask, b.cursor, rec, bk_lb));
    } else {
        hipLaunchKernelGGL((k_ws_bucket<WsEntryPred<MT>, true>), dim3((unsigned)cdiv(g.n, 256 * BK_CH)), dim3(256), (size_t)bk_lb * 4, st, g.n, b.C, WsEntryPred<MT>{b.comp, mk}, b.cursor, b.elist, bk_lb);
    }
    IVX_LAUNCH_CHECK();

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
        const bool links = rec != nullptr && c > 0; // (level 0 = the markers: listed by k_ws_marker_list, no links)
        if (c > 0) {
            if (links) {
                WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_keys_links<CC>, dim3(gb), dim3(256), 0, st, g, b.pmask, b.comp, b.tau, rec, b.key, b.used, start,
                                                          cnt, b.st));
            } else {
                WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_keys<CC>, dim3(gb), dim3(256), 0, st, g, b.pmask, b.comp, b.tau, b.elist, b.key,
                                                          b.used, start, cnt, b.st));
            }
            IVX_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_ws_rank, dim3(1), dim3(1024), 0, st, b.st, b.used, b.remap, b.lab, cap);
        IVX_LAUNCH_CHECK();
        if (links) {
            WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_claim_links<CC>, dim3(gb), dim3(256), 0, st, g, b.zmask, b.comp, b.tau, rec, b.key, b.remap, start,
                                                      cnt));
        } else {
            WS_CONN_SWITCH(conn, hipLaunchKernelGGL(k_ws_claim<CC>, dim3(gb), dim3(256), 0, st, g, b.zmask, b.comp, b.tau, b.elist, b.key,
                                                      b.remap, start, cnt));
        }
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
        stats[13] = levels_done; stats[14] = level_rounds; stats[15] = level_voxels; // the cost map's bit-plane levels
    }
    { // the link records stay cached between floods unless they hold more than an eighth of the device (then: allocated per flood)
        static size_t keep = 0;
        if (!keep) {
            size_t fr = 0, tot = 0;
            keep = hipMemGetInfo(&fr, &tot) == hipSuccess && tot ? tot / 8 : ((size_t)4 << 30);
        }
        const int rc = ws_release_s(WS_WSLINK, st, keep);
        if (rc != IVX_OK) return rc;
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

