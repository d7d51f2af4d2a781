// ivx_runtime.hip -- runtime plumbing of libivx.so: errors, device memory, streams, events,
// cached workspaces and the strided host<->dense device staging used by the host-level entry points.
#include <execinfo.h>
#include <signal.h>
#include <stdarg.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <unistd.h>

#include <map>
#include <vector>
#include <mutex>
#include <thread>
#include <vector>

#include "ivx_internal.h"

namespace ivx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool trace_enabled() {
    static const bool on = getenv("IVX_TRACE") != nullptr;
    return on;
}

struct Slot {
    void *p = nullptr;
    size_t n = 0;
};
static Slot g_ws[WS_COUNT];
static Slot g_hs[WS_COUNT];
static std::mutex g_mu;

int ws_get(int slot, size_t nbytes, void **dptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    Slot &s = g_ws[slot];
    if (nbytes == 0) nbytes = 16;
    if (s.n < nbytes) {
        if (s.p) IVX_HIP(hipFree(s.p));
        s.p = nullptr;
        s.n = 0;
        size_t want = nbytes + (nbytes >> 3) + 4096; // a little slack so growing volumes do not thrash
        IVX_HIP(hipMalloc(&s.p, want));
        s.n = want;
    }
    *dptr = s.p;
    return IVX_OK;
}

static std::map<std::pair<int, hipStream_t>, Slot> g_ws_stream;

int ws_get_s(int slot, hipStream_t stream, size_t nbytes, void **dptr) {
    if (stream == nullptr) return ws_get(slot, nbytes, dptr);
    std::lock_guard<std::mutex> lk(g_mu);
    Slot &s = g_ws_stream[std::make_pair(slot, stream)];
    if (nbytes == 0) nbytes = 16;
    if (s.n < nbytes) {
        if (s.p) IVX_HIP(hipFree(s.p)); // synchronises the device: nothing can still be using the old block
        s.p = nullptr;
        s.n = 0;
        size_t want = nbytes + (nbytes >> 3) + 4096;
        IVX_HIP(hipMalloc(&s.p, want));
        s.n = want;
    }
    *dptr = s.p;
    return IVX_OK;
}

// give a stream's slot back when it holds more than `keep_below` bytes (big, rarely needed blocks: the IFT flood's link records)
int ws_release_s(int slot, hipStream_t stream, size_t keep_below) {
    std::lock_guard<std::mutex> lk(g_mu);
    Slot *s = nullptr;
    if (stream == nullptr) {
        s = &g_ws[slot];
    } else {
        auto it = g_ws_stream.find(std::make_pair(slot, stream));
        if (it == g_ws_stream.end()) return IVX_OK;
        s = &it->second;
    }
    if (s->p && s->n > keep_below) {
        IVX_HIP(hipFree(s->p));
        s->p = nullptr;
        s->n = 0;
    }
    return IVX_OK;
}

static std::recursive_mutex g_host_mu;
static int g_host_depth = 0;       // nesting of host-level entry points (they call each other)
static uint64_t g_host_epoch = 1;  // one per OUTERMOST host call: what upload_strided staged stays valid inside it
static void ivx_bt_handler(int sig);
HostCallGuard::HostCallGuard() {
    g_host_mu.lock();
    if (g_host_depth++ == 0) g_host_epoch++;
    static const bool bt = getenv("IVX_BACKTRACE") != nullptr;
    if (bt) { // re-arm on every host call: test runners install their own handlers after the library is loaded
        signal(SIGABRT, ivx_bt_handler);
        signal(SIGSEGV, ivx_bt_handler);
    }
}
uint64_t host_epoch() { return g_host_epoch; } // (the number of the running outermost host call)
HostCallGuard::~HostCallGuard() {
    g_host_depth--;
    g_host_mu.unlock();
}

int hs_get(int slot, size_t nbytes, void **hptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    Slot &s = g_hs[slot];
    if (nbytes == 0) nbytes = 16;
    if (s.n < nbytes) {
        if (s.p) IVX_HIP(hipHostFree(s.p));
        s.p = nullptr;
        s.n = 0;
        size_t want = nbytes + (nbytes >> 3) + 4096;
        IVX_HIP(hipHostMalloc(&s.p, want, hipHostMallocDefault));
        s.n = want;
    }
    *hptr = s.p;
    return IVX_OK;
}

// ---- pageable host memory at (nearly) the link's rate ----------------------------------------------------------------
// The reference hands np.memmap / plain numpy arrays over (invesalius/data/mask.py:422-431, slice_.py:192): pageable memory.
// What that costs was measured per direction (see copy_h2d / copy_d2h below): a warm pageable upload already runs at the
// link's rate (the runtime locks the user pages), a download into pages that were never touched -- the np.empty result array
// of every call -- does not: one thread takes every page fault and the copy crawls at 25 GB/s.  For that case a copy of
// >= 4 MB is cut into chunks that a few host threads ("lanes") move through their own page-locked double buffers on their own
// streams: while the DMA engine carries chunk k of a lane, the lane's thread unpacks chunk k - lanes, so the page faults of the
// fresh array are spread over the lanes (46 GB/s).  A pointer that is already page-locked (ivx_host_alloc, hipHostRegister)
// goes straight to hipMemcpy.  Synchronous like hipMemcpy: the bytes are in place on return.
namespace {
constexpr size_t STAGE_MIN = 4u << 20;
constexpr int STAGE_MAX_LANES = 16;
static size_t stage_chunk() { // bytes per lane buffer (IVX_STAGE_CHUNK_MB: 1 .. 64, default 4)
    static const size_t n = []() {
        const char *e = getenv("IVX_STAGE_CHUNK_MB");
        long v = e ? atol(e) : 4;
        if (v < 1) v = 1;
        if (v > 64) v = 64;
        return (size_t)v << 20;
    }();
    return n;
}
struct StageLane {
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
};
struct StageCtx {
    int nl = 0;
    StageLane lane[STAGE_MAX_LANES];
};
static std::mutex g_stage_mu; // one staged copy at a time (the lanes are the parallelism)
static std::map<int, StageCtx> g_stage;

static int stage_lanes() {
    static const int n = []() {
        const char *e = getenv("IVX_STAGE_THREADS");
        int v = e ? atoi(e) : 0;
        if (!e) {
            const unsigned hc = std::thread::hardware_concurrency();
            v = hc >= 16 ? 8 : hc >= 8 ? 4 : hc >= 4 ? 2 : 0;
        }
        return v < 0 ? 0 : v > STAGE_MAX_LANES ? STAGE_MAX_LANES : v;
    }();
    return n;
}

static bool host_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // unregistered memory: the query fails, and that failure must not linger
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

static int stage_ctx(int dev, StageCtx **out) {
    StageCtx &c = g_stage[dev];
    const int want = stage_lanes();
    while (c.nl < want) {
        StageLane &l = c.lane[c.nl];
        IVX_HIP(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
        for (int q = 0; q < 2; q++) {
            IVX_HIP(hipHostMalloc(&l.buf[q], stage_chunk(), hipHostMallocDefault));
            IVX_HIP(hipEventCreateWithFlags(&l.ev[q], hipEventDisableTiming));
        }
        c.nl++;
    }
    *out = &c;
    return IVX_OK;
}

// to_device: dev <- host, else host <- dev
static int staged_copy(void *dev_p, void *host_p, size_t n, bool to_device) {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    int dev = 0;
    IVX_HIP(hipGetDevice(&dev));
    StageCtx *c;
    int rc = stage_ctx(dev, &c);
    if (rc) return rc;
    IVX_HIP(hipStreamSynchronize(nullptr));
    const size_t STAGE_CHUNK = stage_chunk();
    const size_t nchunks = (n + STAGE_CHUNK - 1) / STAGE_CHUNK;
    const int nl = (size_t)c->nl < nchunks ? c->nl : (int)nchunks;
    std::vector<hipError_t> err((size_t)nl, hipSuccess);
    auto work = [&](int t) {
        hipError_t e = hipSetDevice(dev);
        StageLane &l = c->lane[t];
        char *d = (char *)dev_p, *h = (char *)host_p;
        size_t prev = 0, prev_len = 0;
        int k = 0;
        for (size_t i = (size_t)t; i < nchunks && e == hipSuccess; i += (size_t)nl, k++) {
            const int q = k & 1;
            const size_t off = i * STAGE_CHUNK, len = n - off < STAGE_CHUNK ? n - off : STAGE_CHUNK;
            if (to_device) {
                if (k >= 2) e = hipEventSynchronize(l.ev[q]); // the DMA that last read this buffer
                if (e != hipSuccess) break;
                memcpy(l.buf[q], h + off, len);
                e = hipMemcpyAsync(d + off, l.buf[q], len, hipMemcpyHostToDevice, l.st);
                if (e == hipSuccess) e = hipEventRecord(l.ev[q], l.st);
            } else {
                e = hipMemcpyAsync(l.buf[q], d + off, len, hipMemcpyDeviceToHost, l.st);
                if (e == hipSuccess) e = hipEventRecord(l.ev[q], l.st);
                if (k >= 1 && e == hipSuccess) { // unpack the previous chunk while this one is in flight
                    e = hipEventSynchronize(l.ev[q ^ 1]);
                    if (e == hipSuccess) memcpy(h + prev, l.buf[q ^ 1], prev_len);
                }
                prev = off;
                prev_len = len;
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(l.st);
        if (!to_device && e == hipSuccess && k >= 1) memcpy(h + prev, l.buf[(k - 1) & 1], prev_len);
        err[(size_t)t] = e;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nl; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    for (int t = 0; t < nl; t++)
        if (err[(size_t)t] != hipSuccess) {
            set_error("staged %s copy of %zu bytes: %s", to_device ? "host-to-device" : "device-to-host", n, hipGetErrorString(err[(size_t)t]));
            return err[(size_t)t] == hipErrorOutOfMemory ? IVX_ENOMEM : IVX_EHIP;
        }
    return IVX_OK;
}
} // namespace

// fraction of a sample of the range's pages that are not resident yet (a fresh np.empty / a new memmap): writing into such a
// range is bound by its page faults, which the lanes take in parallel
static bool mostly_untouched(const void *p, size_t n) {
    const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
    const uintptr_t a0 = ((uintptr_t)p + pg - 1) & ~(uintptr_t)(pg - 1), a1 = ((uintptr_t)p + n) & ~(uintptr_t)(pg - 1);
    if (a1 <= a0 + 32 * pg) return false;
    const size_t npages = (a1 - a0) / pg;
    int missing = 0;
    constexpr int SAMPLES = 32;
    for (int q = 0; q < SAMPLES; q++) {
        unsigned char v = 0;
        const uintptr_t a = a0 + (npages - 1) * (size_t)q / (SAMPLES - 1) * pg;
        if (mincore((void *)a, pg, &v) != 0) return false;
        missing += !(v & 1);
    }
    return missing * 2 > SAMPLES;
}

// Measured on the MI355X box (tools/bench_stage.py, profiles/r04_stage_copies.txt; 630 MB of one bench step):
//   host -> device, pageable, warm: hipMemcpy 55.7 GB/s (the runtime locks the user pages itself), lanes 22 - 50 GB/s  -> hipMemcpy
//   device -> host into pages never touched (np.empty): hipMemcpy 24.9 GB/s (faults taken by one thread), lanes 46 GB/s -> lanes
//   device -> host into touched pages: hipMemcpy 55 GB/s, lanes 47 GB/s                                                 -> hipMemcpy
// IVX_STAGE_THREADS=0 never uses the lanes; IVX_STAGE_UP=1 sends uploads through them as well (A/B).
int copy_h2d(void *dst_dev, const void *src, size_t n) {
    if (!n) return IVX_OK;
    static const bool up = []() { const char *e = getenv("IVX_STAGE_UP"); return e && e[0] == '1'; }();
    if (up && n >= STAGE_MIN && stage_lanes() > 0 && !host_is_pinned(src)) return staged_copy(dst_dev, const_cast<void *>(src), n, true);
    IVX_HIP(hipMemcpy(dst_dev, src, n, hipMemcpyHostToDevice));
    return IVX_OK;
}
int copy_d2h(void *dst, const void *src_dev, size_t n) {
    if (!n) return IVX_OK;
    // IVX_D2H_LANES (read per call: A/B runs flip it inside one process): 1 = every pageable destination through the lanes,
    // 0 = never, unset = where the destination's pages are mostly not resident yet
    const char *force = getenv("IVX_D2H_LANES");
    const bool lanes_ok = n >= STAGE_MIN && stage_lanes() > 0 && !(force && force[0] == '0');
    if (lanes_ok && ((force && force[0] == '1') || mostly_untouched(dst, n)) && !host_is_pinned(dst))
        return staged_copy(const_cast<void *>(src_dev), dst, n, false);
    IVX_HIP(hipMemcpy(dst, src_dev, n, hipMemcpyDeviceToHost));
    return IVX_OK;
}

static inline bool dense3(const int64_t shape[3], const int64_t st[3], size_t isz) {
    return st[2] == (int64_t)isz && st[1] == shape[2] * (int64_t)isz && st[0] == shape[1] * shape[2] * (int64_t)isz;
}

// ---- strided host views at PCIe speed ------------------------------------------------------------------------
// The views the GUI passes -- `mask.matrix[1:, 1:, 1:]` of the (d+1, h+1, w+1) memmap, axis-sliced slabs -- are
// sub-boxes of a C-contiguous parent: contiguous rows, constant pitches.  Such a view lies inside ONE contiguous byte
// span of the parent that is barely larger than the view itself, so it travels as one dense copy each way and the
// re-pitching is a device kernel (TB/s) instead of a host memcpy per row (the round-1 path: 65 ms for a 512^3 mask).
// On the way back the span is first fetched (unless this very call uploaded it), the kernel scatters the result into
// the view's bytes, and the whole span returns: the parent's pad / flag cells between the rows come back unchanged.
struct SpanSlot {
    void *d = nullptr;
    size_t cap = 0;
    const void *host = nullptr;
    size_t nbytes = 0;
    uint64_t epoch = 0;
};
static SpanSlot g_span[WS_COUNT];

static bool span_of(const int64_t shape[3], const int64_t st[3], size_t isz, size_t *span) {
    const int64_t row = shape[2] * (int64_t)isz;
    if (st[2] != (int64_t)isz || st[1] < row || st[0] < st[1] * shape[1]) return false;
    const size_t n = (size_t)shape[0] * shape[1] * row;
    const size_t sp = (size_t)(shape[0] - 1) * st[0] + (size_t)(shape[1] - 1) * st[1] + row;
    if (sp > n + n / 4 + 65536) return false; // a thin slice of a big parent: not worth moving the parent
    *span = sp;
    return true;
}

// dense (rows packed) <-> pitched, 16 bytes of the dense side per lane; TO_PITCHED: dense -> pitched
template <bool TO_PITCHED>
__global__ __launch_bounds__(256) void k_repitch(uint8_t *__restrict__ pitched, int64_t p0, int64_t p1, uint8_t *__restrict__ dense,
                                                 int64_t nz, int64_t ny, int64_t row) {
    const int64_t chunks = (row + 15) / 16;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nz * ny * chunks) return;
    const int64_t c = i % chunks, r = i / chunks, y = r % ny, z = r / ny;
    const int64_t off = c * 16, len = row - off < 16 ? row - off : 16;
    uint8_t *pp = pitched + z * p0 + y * p1 + off;
    uint8_t *dp = dense + (z * ny + y) * row + off;
    uint8_t v[16];
    if (TO_PITCHED) {
        if (len == 16 && ((uintptr_t)dp & 15) == 0) *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(dp);
        else
            for (int q = 0; q < len; q++) v[q] = dp[q];
        if (len == 16 && ((uintptr_t)pp & 15) == 0) *reinterpret_cast<uint4 *>(pp) = *reinterpret_cast<const uint4 *>(v);
        else
            for (int q = 0; q < len; q++) pp[q] = v[q];
    } else {
        if (len == 16 && ((uintptr_t)pp & 15) == 0) *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(pp);
        else
            for (int q = 0; q < len; q++) v[q] = pp[q];
        if (len == 16 && ((uintptr_t)dp & 15) == 0) *reinterpret_cast<uint4 *>(dp) = *reinterpret_cast<const uint4 *>(v);
        else
            for (int q = 0; q < len; q++) dp[q] = v[q];
    }
}

static int span_buffer(int hslot, size_t span, SpanSlot **out) {
    SpanSlot &sp = g_span[hslot];
    if (sp.cap < span) {
        if (sp.d) IVX_HIP(hipFree(sp.d));
        sp.d = nullptr;
        sp.cap = 0;
        const size_t want = span + (span >> 3) + 4096;
        IVX_HIP(hipMalloc(&sp.d, want));
        sp.cap = want;
        sp.epoch = 0;
    }
    *out = &sp;
    return IVX_OK;
}

// rows gathered / scattered by a few host threads (views that are not sub-boxes of a contiguous parent)
template <typename F> static void rows_parallel(int64_t nrows, F f) {
    const int nt = nrows > 4096 ? 4 : 1;
    if (nt == 1) { f(0, nrows); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(f, nrows * t / nt, nrows * (t + 1) / nt);
    for (auto &t : th) t.join();
}

int upload_strided(void *dst_dev, const void *src, const int64_t shape[3], const int64_t st[3], size_t isz,
                   int hslot) {
    const size_t n = (size_t)shape[0] * shape[1] * shape[2] * isz;
    if (n == 0) return IVX_OK;
    if (dense3(shape, st, isz)) return copy_h2d(dst_dev, src, n);
    size_t span = 0;
    if (hslot >= 0 && hslot < WS_COUNT && span_of(shape, st, isz, &span)) {
        SpanSlot *sp;
        int rc = span_buffer(hslot, span, &sp);
        if (rc) return rc;
        if ((rc = copy_h2d(sp->d, src, span))) return rc;
        sp->host = src;
        sp->nbytes = span;
        sp->epoch = g_host_depth > 0 ? g_host_epoch : 0;
        const int64_t row = shape[2] * (int64_t)isz, chunks = (row + 15) / 16;
        hipLaunchKernelGGL(k_repitch<false>, dim3((unsigned)cdiv(shape[0] * shape[1] * chunks, 256)), dim3(256), 0, 0,
                           (uint8_t *)sp->d, st[0], st[1], (uint8_t *)dst_dev, shape[0], shape[1], row);
        IVX_LAUNCH_CHECK();
        return IVX_OK;
    }
    void *h;
    int rc = hs_get(hslot, n, &h);
    if (rc) return rc;
    char *d = (char *)h;
    const char *s = (const char *)src;
    const size_t row = (size_t)shape[2] * isz;
    const int64_t ny = shape[1], nx = shape[2];
    rows_parallel(shape[0] * shape[1], [=](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; r++) {
            const int64_t z = r / ny, y = r - z * ny;
            const char *sp = s + z * st[0] + y * st[1];
            char *dp = d + (size_t)r * row;
            if (st[2] == (int64_t)isz) memcpy(dp, sp, row);
            else
                for (int64_t x = 0; x < nx; x++) memcpy(dp + x * isz, sp + x * st[2], isz);
        }
    });
    IVX_HIP(hipMemcpy(dst_dev, h, n, hipMemcpyHostToDevice));
    return IVX_OK;
}

int download_strided(void *dst, const int64_t shape[3], const int64_t st[3], const void *src_dev, size_t isz,
                     int hslot) {
    const size_t n = (size_t)shape[0] * shape[1] * shape[2] * isz;
    if (n == 0) return IVX_OK;
    if (dense3(shape, st, isz)) return copy_d2h(dst, src_dev, n);
    size_t span = 0;
    if (hslot >= 0 && hslot < WS_COUNT && span_of(shape, st, isz, &span)) {
        SpanSlot *sp;
        int rc = span_buffer(hslot, span, &sp);
        if (rc) return rc;
        const bool fresh = sp->host == dst && sp->nbytes == span && sp->epoch != 0 && sp->epoch == g_host_epoch && g_host_depth > 0;
        if (!fresh && (rc = copy_h2d(sp->d, dst, span))) return rc; // the bytes between the view's rows
        const int64_t row = shape[2] * (int64_t)isz, chunks = (row + 15) / 16;
        hipLaunchKernelGGL(k_repitch<true>, dim3((unsigned)cdiv(shape[0] * shape[1] * chunks, 256)), dim3(256), 0, 0,
                           (uint8_t *)sp->d, st[0], st[1], (uint8_t *)const_cast<void *>(src_dev), shape[0], shape[1], row);
        IVX_LAUNCH_CHECK();
        IVX_HIP(hipStreamSynchronize(nullptr)); // (the kernel above runs on the null stream; the staged copy on its own streams)
        if ((rc = copy_d2h(dst, sp->d, span))) return rc;
        sp->epoch = 0; // the host copy is the truth again
        return IVX_OK;
    }
    void *h;
    int rc = hs_get(hslot, n, &h);
    if (rc) return rc;
    IVX_HIP(hipMemcpy(h, src_dev, n, hipMemcpyDeviceToHost));
    const char *s = (const char *)h;
    char *d = (char *)dst;
    const size_t row = (size_t)shape[2] * isz;
    const int64_t ny = shape[1], nx = shape[2];
    rows_parallel(shape[0] * shape[1], [=](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; r++) {
            const int64_t z = r / ny, y = r - z * ny;
            char *dp = d + z * st[0] + y * st[1];
            const char *sp = s + (size_t)r * row;
            if (st[2] == (int64_t)isz) memcpy(dp, sp, row);
            else
                for (int64_t x = 0; x < nx; x++) memcpy(dp + x * st[2], sp + x * isz, isz);
        }
    });
    return IVX_OK;
}

int download_strided2(void *dst, const int64_t shape[2], const int64_t st[2], const void *src_dev, size_t isz,
                      int hslot) {
    const int64_t sh3[3] = {1, shape[0], shape[1]};
    const int64_t st3[3] = {0, st[0], st[1]};
    if (st[1] == (int64_t)isz && st[0] == shape[1] * (int64_t)isz) {
        const size_t n = (size_t)shape[0] * shape[1] * isz;
        return copy_d2h(dst, src_dev, n);
    }
    return download_strided(dst, sh3, st3, src_dev, isz, hslot);
}

// ---- mailbox ------------------------------------------------------------------------------------------------
static uint32_t *g_mb = nullptr; // pinned host memory, 64 dwords per slot x 64 slots (ring)
static uint32_t g_mb_seq = 0;

__global__ void k_mailbox_publish(const uint32_t *__restrict__ src, int n, uint32_t *mb, uint32_t seq) {
    if (threadIdx.x || blockIdx.x) return;
    for (int i = 0; i < n; i++) mb[i] = src[i];
    __threadfence_system();
    __hip_atomic_store(&mb[63], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int mailbox_publish(const void *dsrc, int ndwords, hipStream_t st, uint32_t *seq_out) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_mb) {
            void *p = nullptr;
            IVX_HIP(hipHostMalloc(&p, 64 * 64 * 4, hipHostMallocMapped | hipHostMallocCoherent));
            memset(p, 0, 64 * 64 * 4);
            g_mb = (uint32_t *)p;
        }
        *seq_out = ++g_mb_seq;
        if (*seq_out == 0) *seq_out = ++g_mb_seq;
    }
    IVX_REQUIRE(ndwords >= 0 && ndwords <= 32, IVX_EINVAL, "mailbox: too many words");
    uint32_t *slot = g_mb + (size_t)(*seq_out & 63u) * 64;
    hipLaunchKernelGGL(k_mailbox_publish, dim3(1), dim3(64), 0, st, (const uint32_t *)dsrc, ndwords, slot, *seq_out);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// a slot and a sequence number for a kernel that publishes by itself (dwords 0 .. n-1, a system-scope fence, then the sequence
// number into dword 63 with release semantics at system scope -- what k_mailbox_publish does): saves the publishing launch
int mailbox_reserve(uint32_t **slot_out, uint32_t *seq_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_mb) {
        void *p = nullptr;
        IVX_HIP(hipHostMalloc(&p, 64 * 64 * 4, hipHostMallocMapped | hipHostMallocCoherent));
        memset(p, 0, 64 * 64 * 4);
        g_mb = (uint32_t *)p;
    }
    *seq_out = ++g_mb_seq;
    if (*seq_out == 0) *seq_out = ++g_mb_seq;
    *slot_out = g_mb + (size_t)(*seq_out & 63u) * 64;
    return IVX_OK;
}

// the sharded flood's vote (parallel.py, slab_region_grow): two device words { everybody's votes of the last round, my new count }.
// Setting them and reading them used to be two memsets, a 4-byte device copy and a synchronising 8-byte download per round --
// 16 us each as copy-engine calls; here: one-thread kernels and the mailbox (no stream synchronisation).
__global__ void k_vote_set(int32_t *v, int32_t a, int32_t b) {
    v[0] = a;
    v[1] = b;
}
// publish both words, then stage the next round's vote (votes[0] <- my count: the next exchange all-reduces it in place)
__global__ void k_vote_publish(int32_t *v, uint32_t *mb, uint32_t seq) {
    if (threadIdx.x || blockIdx.x) return;
    const int32_t a = v[0], b = v[1];
    mb[0] = (uint32_t)a;
    mb[1] = (uint32_t)b;
    v[0] = b;
    v[1] = 0; // (ivx_dev_flood_or_planes_acc adds the next round's gains to it)
    __threadfence_system();
    __hip_atomic_store(&mb[63], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int vote_publish(int32_t *votes, hipStream_t st, uint32_t *seq_out) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_mb) {
            void *p = nullptr;
            IVX_HIP(hipHostMalloc(&p, 64 * 64 * 4, hipHostMallocMapped | hipHostMallocCoherent));
            memset(p, 0, 64 * 64 * 4);
            g_mb = (uint32_t *)p;
        }
        *seq_out = ++g_mb_seq;
        if (*seq_out == 0) *seq_out = ++g_mb_seq;
    }
    uint32_t *slot = g_mb + (size_t)(*seq_out & 63u) * 64;
    hipLaunchKernelGGL(k_vote_publish, dim3(1), dim3(64), 0, st, votes, slot, *seq_out);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

int mailbox_wait(uint32_t seq, hipStream_t st, uint32_t *out, int ndwords) {
    volatile uint32_t *slot = g_mb + (size_t)(seq & 63u) * 64;
    bool ok = false;
    for (long spins = 0; spins < 20000000L; spins++) { // ~ a few hundred ms worst case, then the safe path
        if (__atomic_load_n(&slot[63], __ATOMIC_ACQUIRE) == seq) { ok = true; break; }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!ok) {
        IVX_HIP(hipStreamSynchronize(st));
        IVX_REQUIRE(__atomic_load_n(&slot[63], __ATOMIC_ACQUIRE) == seq, IVX_EHIP, "mailbox: GPU never published sequence %u", seq);
    }
    for (int i = 0; i < ndwords; i++) out[i] = slot[i];
    return IVX_OK;
}

// ---- progress lines ----------------------------------------------------------------------------------------
// one 64-B line per stream, carved from pinned chunks of 64 lines; a line is handed back when its stream is destroyed
struct ProgressLine {
    unsigned long long *p;
    uint8_t tag;
};
static std::map<hipStream_t, ProgressLine> g_pl;
static std::vector<unsigned long long *> g_pl_free;

int progress_line(hipStream_t st, volatile unsigned long long **line, uint32_t *tag) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_pl.find(st);
    if (it == g_pl.end()) {
        if (g_pl_free.empty()) {
            void *p = nullptr;
            IVX_HIP(hipHostMalloc(&p, 64 * 64, hipHostMallocMapped | hipHostMallocCoherent));
            memset(p, 0, 64 * 64);
            for (int i = 63; i >= 0; i--) g_pl_free.push_back((unsigned long long *)p + (size_t)i * 8);
        }
        ProgressLine l = {g_pl_free.back(), 0};
        g_pl_free.pop_back();
        it = g_pl.insert(std::make_pair(st, l)).first;
    }
    if (++it->second.tag == 0) it->second.tag = 1;
    *tag = it->second.tag;
    *line = it->second.p;
    return IVX_OK;
}

void progress_forget_stream(void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_pl.find((hipStream_t)stream);
    if (it == g_pl.end()) return;
    // the stream has been synchronised by the caller: nothing can still write to the line.  The next owner starts from
    // a clean line, so the tag may start over.
    memset(it->second.p, 0, 64);
    g_pl_free.push_back(it->second.p);
    g_pl.erase(it);
}

} // namespace ivx

using namespace ivx;

// IVX_BACKTRACE=1: print a native backtrace on SIGABRT / SIGSEGV (diagnostic aid; off by default)
static void ivx::ivx_bt_handler(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "ivx: fatal signal, native backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void ivx_install_bt() {
    if (getenv("IVX_BACKTRACE")) {
        signal(SIGABRT, ivx_bt_handler);
        signal(SIGSEGV, ivx_bt_handler);
    }
}

extern "C" {

int ivx_version(void) { return 100; }
const char *ivx_last_error(void) { return g_err; }

int ivx_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        n = 0;
        (void)hipGetLastError();
    }
    *count = n;
    return IVX_OK;
}
int ivx_set_device(int device) {
    IVX_HIP(hipSetDevice(device));
    return IVX_OK;
}
int ivx_device_synchronize(void) {
    IVX_HIP(hipDeviceSynchronize());
    return IVX_OK;
}
int ivx_device_name(char *buf, size_t buflen) {
    int dev = 0;
    IVX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    IVX_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return IVX_OK;
}
int ivx_malloc(void **dptr, size_t nbytes) {
    IVX_HIP(hipMalloc(dptr, nbytes ? nbytes : 16));
    return IVX_OK;
}
int ivx_free(void *dptr) {
    if (dptr) IVX_HIP(hipFree(dptr));
    return IVX_OK;
}
int ivx_memset(void *dptr, int value, size_t nbytes, void *stream) {
    if (nbytes) IVX_HIP(hipMemsetAsync(dptr, value, nbytes, S(stream)));
    return IVX_OK;
}
// (pageable host memory of >= 4 MB travels through the page-locked lane buffers above; IVX_STAGE_THREADS=0: plain hipMemcpy)
int ivx_memcpy_h2d(void *dst, const void *src, size_t nbytes) { return copy_h2d(dst, src, nbytes); }
int ivx_memcpy_d2h(void *dst, const void *src, size_t nbytes) { return copy_d2h(dst, src, nbytes); }
// Page-locked host memory for callers that keep their arrays where the DMA engines can reach them directly: a pageable
// numpy array crosses PCIe through the runtime's bounce buffers (~25-40 GB/s here), a pinned one at the link's rate.
int ivx_host_alloc(void **hptr, size_t nbytes) {
    IVX_REQUIRE(hptr, IVX_EINVAL, "ivx_host_alloc: null");
    IVX_HIP(hipHostMalloc(hptr, nbytes ? nbytes : 16, hipHostMallocDefault));
    return IVX_OK;
}
int ivx_host_free(void *hptr) {
    if (hptr) IVX_HIP(hipHostFree(hptr));
    return IVX_OK;
}
int ivx_memcpy_d2d(void *dst, const void *src, size_t nbytes, void *stream) {
    if (nbytes) IVX_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, S(stream)));
    return IVX_OK;
}
int ivx_dev_vote_set(int32_t *votes, int32_t v0, int32_t v1, void *stream) {
    IVX_REQUIRE(votes, IVX_EINVAL, "ivx_dev_vote_set: null");
    hipLaunchKernelGGL(ivx::k_vote_set, dim3(1), dim3(1), 0, S(stream), votes, v0, v1);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
int ivx_dev_vote_read(int32_t *votes, int32_t out[2], void *stream) {
    IVX_REQUIRE(votes && out, IVX_EINVAL, "ivx_dev_vote_read: null");
    uint32_t seq = 0, w[2] = {0, 0};
    int rc = ivx::vote_publish(votes, S(stream), &seq);
    if (rc != IVX_OK) return rc;
    rc = ivx::mailbox_wait(seq, S(stream), w, 2);
    if (rc != IVX_OK) return rc;
    out[0] = (int32_t)w[0];
    out[1] = (int32_t)w[1];
    return IVX_OK;
}
int ivx_stream_create(void **stream) {
    hipStream_t s;
    IVX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return IVX_OK;
}
// a stream whose work yields to the default-priority streams' when both have workgroups waiting (background work that
// should fill idle CUs without slowing the main stream down)
int ivx_stream_create_low_priority(void **stream) {
    int least = 0, greatest = 0;
    IVX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s;
    IVX_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least));
    *stream = (void *)s;
    return IVX_OK;
}
int ivx_stream_destroy(void *stream) {
    if (!stream) return IVX_OK;
    IVX_HIP(hipStreamSynchronize(S(stream)));
    { // the stream's private workspaces go with it (a later stream may be handed the same handle value)
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto it = g_ws_stream.begin(); it != g_ws_stream.end();) {
            if (it->first.second == S(stream)) {
                if (it->second.p) (void)hipFree(it->second.p);
                it = g_ws_stream.erase(it);
            } else {
                ++it;
            }
        }
    }
    ccl_forget_stream(stream);
    progress_forget_stream(stream);
    IVX_HIP(hipStreamDestroy(S(stream)));
    return IVX_OK;
}
int ivx_stream_synchronize(void *stream) {
    IVX_HIP(hipStreamSynchronize(S(stream)));
    return IVX_OK;
}
int ivx_event_create(void **event) {
    hipEvent_t e;
    // timing events between kernels of one stream: no system-scope fence when the event fires (a default event drains
    // the caches to host visibility and leaves a ~5 us bubble in the stream every time it is recorded)
    IVX_HIP(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    *event = (void *)e;
    return IVX_OK;
}
// an event for ORDERING streams (ivx_event_record + ivx_stream_wait_event), not for timing: full fence, no timestamps
int ivx_event_create_sync(void **event) {
    hipEvent_t e;
    IVX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void *)e;
    return IVX_OK;
}
int ivx_event_destroy(void *event) {
    if (event) IVX_HIP(hipEventDestroy((hipEvent_t)event));
    return IVX_OK;
}
int ivx_event_record(void *event, void *stream) {
    IVX_HIP(hipEventRecord((hipEvent_t)event, S(stream)));
    return IVX_OK;
}
int ivx_stream_wait_event(void *stream, void *event) {
    IVX_HIP(hipStreamWaitEvent(S(stream), (hipEvent_t)event, 0));
    return IVX_OK;
}
int ivx_event_elapsed_ms(void *start, void *stop, float *ms) {
    IVX_HIP(hipEventSynchronize((hipEvent_t)stop));
    IVX_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return IVX_OK;
}
int ivx_release_workspace(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < WS_COUNT; i++) {
        if (g_ws[i].p) (void)hipFree(g_ws[i].p);
        g_ws[i] = Slot();
        if (g_hs[i].p) (void)hipHostFree(g_hs[i].p);
        g_hs[i] = Slot();
        if (ivx::g_span[i].d) (void)hipFree(ivx::g_span[i].d);
        ivx::g_span[i] = ivx::SpanSlot();
    }
    for (auto &kv : g_ws_stream)
        if (kv.second.p) (void)hipFree(kv.second.p);
    g_ws_stream.clear();
    return IVX_OK;
}

} // extern "C"
