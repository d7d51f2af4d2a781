// ivx_internal.h -- shared by every translation unit of libivx.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ivx.h"

namespace ivx {

void set_error(const char *fmt, ...);

#define IVX_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            ivx::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return e__ == hipErrorOutOfMemory ? IVX_ENOMEM : IVX_EHIP;                     \
        }                                                                                  \
    } while (0)

// IVX_TRACE=1: synchronise after every kernel launch and log its source line (localises asynchronous GPU faults)
bool trace_enabled();
#define IVX_LAUNCH_CHECK()                                                                 \
    do {                                                                                   \
        IVX_HIP(hipGetLastError());                                                        \
        if (ivx::trace_enabled()) {                                                        \
            fprintf(stderr, "ivx trace: launched %s:%d ...", __FILE__, __LINE__);          \
            fflush(stderr);                                                                \
            IVX_HIP(hipDeviceSynchronize());                                               \
            fprintf(stderr, " ok\n");                                                      \
        }                                                                                  \
    } while (0)

#define IVX_REQUIRE(cond, code, ...)                                                       \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            ivx::set_error(__VA_ARGS__);                                                   \
            return (code);                                                                 \
        }                                                                                  \
    } while (0)

static inline hipStream_t S(void *s) { return (hipStream_t)s; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t dtype_size(int dt) {
    switch (dt) {
    case IVX_U8: return 1;
    case IVX_I16: return 2;
    case IVX_U16: return 2;
    case IVX_F64: return 8;
    default: return 0;
    }
}

// Cached device workspaces for the host-level entry points (grow-only, freed by ivx_release_workspace).
enum { WS_IN = 0, WS_OUT, WS_AUX0, WS_AUX1, WS_AUX2, WS_AUX3, WS_SMALL, WS_CCL0, WS_CCL1, WS_MCLIST, WS_MCV, WS_MCST, WS_MESH, WS_MESH2, WS_HOLES, WS_LUT, WS_WSIFT, WS_WSSK, WS_WSA, WS_WSLINK, WS_COUNT };
int ws_get(int slot, size_t nbytes, void **dptr);
// Same, but private to `stream`: the device-level entry points keep their internal scratch (MC triangle list, MIP
// partials, union-find tables, ...) per stream, so several resident volumes can run concurrently from different
// host threads, each on its own stream.
int ws_get_s(int slot, hipStream_t stream, size_t nbytes, void **dptr);
uint64_t host_epoch(); // number of the running outermost host-level call (HostCallGuard)
int ws_release_s(int slot, hipStream_t stream, size_t keep_below); // frees the slot's block when it is larger than keep_below
// The host-level entry points (ivx_* without _dev_) share the stream-0 workspaces and are serialised by this lock
// (ctypes drops the GIL; the PyO3 originals held it).
struct HostCallGuard {
    HostCallGuard();
    ~HostCallGuard();
};
// pinned host staging (grow-only)
int hs_get(int slot, size_t nbytes, void **hptr);

// dense host <-> device copies, synchronous like hipMemcpy; pageable memory of >= 4 MB goes through page-locked lane buffers
// filled / drained by a few host threads (ivx_runtime.hip)
int copy_h2d(void *dst_dev, const void *src, size_t n);
int copy_d2h(void *dst, const void *src_dev, size_t n);

// strided host <-> dense device helpers (host side gather / scatter + one hipMemcpy)
int upload_strided(void *dst_dev, const void *src, const int64_t shape[3], const int64_t strides[3], size_t isz,
                   int hslot);
int download_strided(void *dst, const int64_t shape[3], const int64_t strides[3], const void *src_dev, size_t isz,
                     int hslot);
int download_strided2(void *dst, const int64_t shape[2], const int64_t strides[2], const void *src_dev, size_t isz,
                      int hslot);

// run-based union-find flood (k_ccl.hip)
void ccl_forget_stream(void *stream);
void ccl_invalidate(const void *scratch);
int ccl_small_components(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *out, uint32_t max_size, int *any_small,
                         const void *scratch_key, hipStream_t st);
bool ccl_supported(uint32_t strct_bits);
int ccl_run(const ivx_flood_plan *p, const uint64_t *cand, uint64_t *reached, const void *scratch_key, hipStream_t st);

// GPU -> host mailbox in pinned, host-coherent memory: a 1-thread kernel copies up to 32 dwords and then stores a
// sequence number; the host spins on the sequence word instead of paying a stream synchronisation round trip
// (~5 us instead of ~40 us).  Falls back to hipStreamSynchronize if the word does not arrive in time.
int mailbox_publish(const void *dsrc, int ndwords, hipStream_t st, uint32_t *seq_out);
int mailbox_wait(uint32_t seq, hipStream_t st, uint32_t *out, int ndwords);
int mailbox_reserve(uint32_t **slot_out, uint32_t *seq_out); // for a kernel that publishes by itself (see ivx_runtime.hip)
int vote_publish(int32_t *votes, hipStream_t st, uint32_t *seq_out); // { votes[0], votes[1] } -> mailbox, then votes[0] <- votes[1]
// Progress line: 64 B of pinned, host-coherent memory per stream that the kernels of a running chain store to directly
// (no publishing kernel) while the host polls it.  Every call hands out a fresh 8-bit tag (never 0) for bits 63..56 of
// the words the chain stores, so values left behind by an earlier chain on the same stream -- kernels that were queued
// ahead and still run after their host loop returned -- are recognised and ignored; on one stream the newest chain
// always writes last.
int progress_line(hipStream_t stream, volatile unsigned long long **line, uint32_t *tag);
void progress_forget_stream(void *stream);

} // namespace ivx
