// k_threshold.hip -- int16 volume -> uint8 mask.
//
// Reference semantics (bit-exact):
//   Slice.do_threshold_to_a_slice   invesalius/data/slice_.py:1722-1737   (PRESERVE: keep 1/2/253/254)
//   Slice.do_threshold_to_all_slices invesalius/data/slice_.py:1739-1769  (per-slice skip flag)
//   Slice.SetMaskThreshold (volume)  invesalius/data/slice_.py:1240-1247  (no preserve rule)
//
// Roofline: pure HBM streaming.  Algorithmic bytes: 2 B read + 1 B written per voxel (3 B/voxel),
// +1 B/voxel for the mask read when PRESERVE.  Each lane moves 32 B in (two 16-B loads) and 16 B out
// (one 16-B store): 64 lanes x 16 B = 1 KiB per wave instruction, fully coalesced; a grid-stride loop over
// <= 2048*8 workgroups keeps every CU's memory pipe full.  No LDS, no MFMA.
#include "ivx_internal.h"

typedef short short8_t __attribute__((ext_vector_type(8)));
typedef unsigned char uchar16_t __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ unsigned char keep_or(unsigned char m, unsigned char v) {
    // existing manual-edit / watershed values survive a re-threshold (slice_.py:1732-1735)
    return (m == 1 || m == 2 || m == 253 || m == 254) ? m : v;
}

template <bool PRESERVE>
__global__ __launch_bounds__(256) void k_threshold16(const short8_t *__restrict__ img, uchar16_t *__restrict__ mask,
                                                     int64_t nchunks, int lo, int hi,
                                                     const uint8_t *__restrict__ skip, int64_t chunks_per_slice) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += stride) {
        if (skip && skip[c / chunks_per_slice]) continue;
        const short8_t a = __builtin_nontemporal_load(&img[2 * c]);
        const short8_t b = __builtin_nontemporal_load(&img[2 * c + 1]);
        uchar16_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            r[i] = ((int)a[i] >= lo && (int)a[i] <= hi) ? 255 : 0;
            r[8 + i] = ((int)b[i] >= lo && (int)b[i] <= hi) ? 255 : 0;
        }
        if (PRESERVE) {
            const uchar16_t m = mask[c];
#pragma unroll
            for (int i = 0; i < 16; i++) r[i] = keep_or(m[i], r[i]);
        }
        mask[c] = r;
    }
}

// threshold + the mask's inside-bit plane (mask >= 127 <=> in range) in one pass: 2 B read, 1 + 1/8 B written per voxel.
// The plane has the layout the region-growing and marching-cubes kernels share (64 x-voxels per uint64, rows whole
// words: requires dx % 64 == 0), so a resident pipeline gets the flood's candidate plane and the surface's inside
// plane as by-products of the pass that reads the image anyway.
__global__ __launch_bounds__(256) void k_threshold16_bits(const short8_t *__restrict__ img, uchar16_t *__restrict__ mask,
                                                          uint16_t *__restrict__ bits, int64_t nchunks, int lo, int hi) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += stride) {
        const short8_t a = __builtin_nontemporal_load(&img[2 * c]);
        const short8_t b = __builtin_nontemporal_load(&img[2 * c + 1]);
        uchar16_t r;
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool in0 = (int)a[i] >= lo && (int)a[i] <= hi, in1 = (int)b[i] >= lo && (int)b[i] <= hi;
            r[i] = in0 ? 255 : 0;
            r[8 + i] = in1 ? 255 : 0;
            m |= (in0 ? 1u : 0u) << i;
            m |= (in1 ? 1u : 0u) << (8 + i);
        }
        mask[c] = r; // (a non-temporal store measured slower for the step: mask[reached] = 254 reads these bytes back)
        bits[c] = (uint16_t)m;
    }
}

// scalar path: tails, unaligned pointers, slices whose size is not a multiple of 16
template <bool PRESERVE>
__global__ __launch_bounds__(256) void k_threshold1(const int16_t *__restrict__ img, uint8_t *__restrict__ mask,
                                                    int64_t begin, int64_t n, int lo, int hi,
                                                    const uint8_t *__restrict__ skip, int64_t slice_elems) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (skip && skip[i / slice_elems]) continue;
        const int v = img[i];
        unsigned char r = (v >= lo && v <= hi) ? 255 : 0;
        if (PRESERVE) r = keep_or(mask[i], r);
        mask[i] = r;
    }
}

} // namespace

extern "C" int ivx_dev_threshold_i16(const int16_t *img, int64_t dz, int64_t dy, int64_t dx, int lo, int hi,
                                     int preserve, const uint8_t *skip_flags, uint8_t *mask, void *stream) {
    IVX_REQUIRE(dz >= 0 && dy >= 0 && dx >= 0, IVX_EINVAL, "threshold: negative shape");
    const int64_t slice = dy * dx, n = dz * slice;
    if (n == 0) return IVX_OK;
    const bool aligned = (((uintptr_t)img | (uintptr_t)mask) & 15) == 0;
    const bool vec_ok = aligned && (!skip_flags || slice % 16 == 0);
    int64_t done = 0;
    if (vec_ok && n >= 16) {
        const int64_t nchunks = n / 16;
        const int64_t blocks = ivx::cdiv(nchunks, 256);
        const int grid = (int)(blocks < 16384 ? blocks : 16384);
        const int64_t cps = skip_flags ? slice / 16 : 1;
        if (preserve)
            hipLaunchKernelGGL(k_threshold16<true>, dim3(grid), dim3(256), 0, ivx::S(stream), (const short8_t *)img,
                               (uchar16_t *)mask, nchunks, lo, hi, skip_flags, cps);
        else
            hipLaunchKernelGGL(k_threshold16<false>, dim3(grid), dim3(256), 0, ivx::S(stream), (const short8_t *)img,
                               (uchar16_t *)mask, nchunks, lo, hi, skip_flags, cps);
        IVX_LAUNCH_CHECK();
        done = nchunks * 16;
    }
    if (done < n) {
        const int64_t rem = n - done;
        const int64_t blocks = ivx::cdiv(rem, 256);
        const int grid = (int)(blocks < 16384 ? blocks : 16384);
        if (preserve)
            hipLaunchKernelGGL(k_threshold1<true>, dim3(grid), dim3(256), 0, ivx::S(stream), img, mask, done, n, lo, hi,
                               skip_flags, slice);
        else
            hipLaunchKernelGGL(k_threshold1<false>, dim3(grid), dim3(256), 0, ivx::S(stream), img, mask, done, n, lo,
                               hi, skip_flags, slice);
        IVX_LAUNCH_CHECK();
    }
    return IVX_OK;
}

extern "C" int ivx_dev_threshold_i16_bits(const int16_t *img, int64_t dz, int64_t dy, int64_t dx, int lo, int hi,
                                          uint8_t *mask, uint64_t *bits, void *stream) {
    IVX_REQUIRE(dz >= 0 && dy >= 0 && dx >= 0, IVX_EINVAL, "threshold: negative shape");
    IVX_REQUIRE(dx % 64 == 0 && (((uintptr_t)img | (uintptr_t)mask | (uintptr_t)bits) & 15) == 0, IVX_EINVAL,
                "threshold_bits: needs dx %% 64 == 0 and 16-byte aligned buffers (use ivx_dev_threshold_i16)");
    const int64_t n = dz * dy * dx;
    if (n == 0) return IVX_OK;
    const int64_t nchunks = n / 16;
    const int64_t blocks = ivx::cdiv(nchunks, 256);
    hipLaunchKernelGGL(k_threshold16_bits, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, ivx::S(stream),
                       (const short8_t *)img, (uchar16_t *)mask, (uint16_t *)bits, nchunks, lo, hi);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

// Host form: see include/ivx.h.  `mask` is the full (dz+1,dy+1,dx+1) matrix.
extern "C" int ivx_threshold_all_slices(const int16_t *img, const int64_t shape[3], const int64_t ist[3], int lo,
                                        int hi, int preserve, int honour_flags, uint8_t *mask,
                                        const int64_t mst[3]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    const int64_t dz = shape[0], dy = shape[1], dx = shape[2];
    IVX_REQUIRE(dz >= 0 && dy >= 0 && dx >= 0, IVX_EINVAL, "threshold: negative shape");
    const size_t n = (size_t)dz * dy * dx;
    if (n == 0) return IVX_OK;
    void *d_img, *d_mask, *d_flags;
    int rc;
    if ((rc = ws_get(WS_IN, n * 2, &d_img))) return rc;
    if ((rc = ws_get(WS_OUT, n, &d_mask))) return rc;
    if ((rc = ws_get(WS_SMALL, (size_t)dz, &d_flags))) return rc;
    if ((rc = upload_strided(d_img, img, shape, ist, 2, WS_IN))) return rc;
    // interior view mask[1:,1:,1:]
    uint8_t *inner = mask + mst[0] + mst[1] + mst[2];
    bool any_skip = false;
    void *hflags;
    if ((rc = hs_get(WS_SMALL, (size_t)dz, &hflags))) return rc;
    uint8_t *hf = (uint8_t *)hflags;
    for (int64_t z = 0; z < dz; z++) {
        hf[z] = honour_flags ? mask[(z + 1) * mst[0]] : 0; // mask.matrix[n,0,0], slice_.py:1761
        any_skip |= hf[z] != 0;
    }
    if (preserve || any_skip) {
        // the preserve rule reads the existing mask; skipped slices must come back unchanged
        if ((rc = upload_strided(d_mask, inner, shape, mst, 1, WS_OUT))) return rc;
    }
    if (any_skip) IVX_HIP(hipMemcpy(d_flags, hf, (size_t)dz, hipMemcpyHostToDevice));
    rc = ivx_dev_threshold_i16((const int16_t *)d_img, dz, dy, dx, lo, hi, preserve,
                               any_skip ? (const uint8_t *)d_flags : nullptr, (uint8_t *)d_mask, nullptr);
    if (rc) return rc;
    IVX_HIP(hipDeviceSynchronize());
    if ((rc = download_strided(inner, shape, mst, d_mask, 1, WS_OUT))) return rc;
    for (int64_t z = 0; z < dz; z++) mask[(z + 1) * mst[0]] = honour_flags && hf[z] ? hf[z] : 1; // slice_.py:1767, 1247
    return IVX_OK;
}
