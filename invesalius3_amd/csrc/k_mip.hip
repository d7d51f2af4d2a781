// k_mip.hip -- MaxIP / MinIP / MeanIP along an axis.
//
// Reference semantics: numpy `.max(axis)`, `.min(axis)`, `.mean(axis)` on the slab in
// Slice.get_image_slice (invesalius/data/slice_.py:885-889 axis 0, 969-973 axis 1, 1056-1060 axis 2).
// max/min keep the image dtype; mean is float64 (numpy sums int16 in float64: every partial sum is an
// integer < 2^53, so the sum is exact in any order and one IEEE division gives numpy's bits).
//
// Roofline: HBM streaming, 2 B/voxel read (int16) + one output pixel per ray.
//   axis 0/1 (rays strided): one lane per group of 8 adjacent output pixels, 16-B loads walking the reduced
//       axis; the reduced axis is split into SPLIT segments (partial images combined by a tiny second kernel)
//       so >= 2048 workgroups are in flight even for a 512x512 output.
//   axis 2 (rays contiguous): one wave per ray, 16-B loads per lane, DPP/shuffle tree at the end.
#include "ivx_internal.h"

namespace {

template <typename T> struct Acc { typedef long long type; };

template <typename T, int OP> struct Red {
    __device__ static __forceinline__ long long init() {
        return OP == IVX_MIP_MAX ? (long long)INT64_MIN : OP == IVX_MIP_MIN ? (long long)INT64_MAX : 0ll;
    }
    __device__ static __forceinline__ long long comb(long long a, long long b) {
        return OP == IVX_MIP_MAX ? (a > b ? a : b) : OP == IVX_MIP_MIN ? (a < b ? a : b) : a + b;
    }
};

template <int OP> struct Red32 {
    __device__ static __forceinline__ int comb(int a, int b) {
        return OP == IVX_MIP_MAX ? (a > b ? a : b) : OP == IVX_MIP_MIN ? (a < b ? a : b) : a + b;
    }
};

// generic strided reduce: out pixel (r, c) = reduce_l vol[r*sr + c*sc + l*sl], c contiguous (sc == 1).
// One lane handles VEC adjacent c; the l range is split across blockIdx.y.
// partial results are int32: a segment is at most 32767 samples of a <= 16-bit type, so |sum| < 2^31
typedef int part_t;
template <typename T, int OP, int VEC>
__global__ __launch_bounds__(256) void k_reduce_strided(const T *__restrict__ vol, int64_t nr, int64_t nc, int64_t len,
                                                        int64_t sr, int64_t sl, int64_t seg,
                                                        part_t *__restrict__ partial) {
    const int64_t ncg = (nc + VEC - 1) / VEC;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nr * ncg) return;
    const int64_t r = t / ncg, cg = t - r * ncg;
    const int64_t c0 = cg * VEC;
    const int64_t l0 = (int64_t)blockIdx.y * seg;
    const int64_t l1 = l0 + seg < len ? l0 + seg : len;
    int acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = OP == IVX_MIP_MAX ? INT32_MIN : OP == IVX_MIP_MIN ? INT32_MAX : 0;
    const T *p = vol + r * sr + c0;
    const bool full = (c0 + VEC <= nc) && ((((uintptr_t)(p + l0 * sl)) | ((uintptr_t)(sl * sizeof(T)))) % (VEC * sizeof(T)) == 0);
    if (full) {
        typedef T vec_t __attribute__((ext_vector_type(VEC)));
#pragma unroll 4
        for (int64_t l = l0; l < l1; l++) {
            const vec_t x = *reinterpret_cast<const vec_t *>(p + l * sl); // (non-temporal loads measured the same: 53 / 55 / 62 us)
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = Red32<OP>::comb(acc[v], (int)x[v]);
        }
    } else {
        for (int64_t l = l0; l < l1; l++)
#pragma unroll
            for (int v = 0; v < VEC; v++)
                if (c0 + v < nc) acc[v] = Red32<OP>::comb(acc[v], (int)p[l * sl + v]);
    }
    part_t *o = partial + ((int64_t)blockIdx.y * nr + r) * nc + c0;
#pragma unroll
    for (int v = 0; v < VEC; v++)
        if (c0 + v < nc) o[v] = acc[v];
}

// contiguous rays: one wave per ray
template <typename T, int OP, int VEC>
__global__ __launch_bounds__(256) void k_reduce_rows(const T *__restrict__ vol, int64_t nrays, int64_t len,
                                                     long long *__restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= nrays) return;
    const T *p = vol + ray * len;
    long long acc = Red<T, OP>::init();
    const bool al = (((uintptr_t)p) % (VEC * sizeof(T))) == 0;
    int64_t x = 0;
    if (al) {
        typedef T vec_t __attribute__((ext_vector_type(VEC)));
        for (x = (int64_t)lane * VEC; x + VEC <= len; x += 64 * VEC) {
            const vec_t v = *reinterpret_cast<const vec_t *>(p + x);
#pragma unroll
            for (int e = 0; e < VEC; e++) acc = Red<T, OP>::comb(acc, (long long)v[e]);
        }
        // x is now the first element this lane could not cover with a full vector; tail below
        const int64_t tail0 = (len / VEC) * VEC;
        for (int64_t y = tail0 + lane; y < len; y += 64) acc = Red<T, OP>::comb(acc, (long long)p[y]);
    } else {
        for (int64_t y = lane; y < len; y += 64) acc = Red<T, OP>::comb(acc, (long long)p[y]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = Red<T, OP>::comb(acc, __shfl_xor(acc, o, 64));
    if (lane == 0) partial[ray] = acc;
}

template <typename T, int OP, typename P>
__global__ __launch_bounds__(256) void k_combine(const P *__restrict__ partial, int64_t npix, int split,
                                                 int64_t len, void *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    long long acc = (long long)partial[i];
    for (int s = 1; s < split; s++) acc = Red<T, OP>::comb(acc, (long long)partial[(int64_t)s * npix + i]);
    if (OP == IVX_MIP_MEAN) ((double *)out)[i] = (double)acc / (double)len;
    else if (OP == IVX_MIP_SUM) ((long long *)out)[i] = acc;
    else ((T *)out)[i] = (T)acc;
}

template <typename T, int OP>
static int run_reduce(const void *vol_, int64_t dz, int64_t dy, int64_t dx, int axis, void *out, hipStream_t st) {
    const T *vol = (const T *)vol_;
    constexpr int VEC = 16 / sizeof(T);
    int64_t nr, nc, len;
    if (axis == 0) { nr = dy; nc = dx; len = dz; }
    else if (axis == 1) { nr = dz; nc = dx; len = dy; }
    else { nr = dz; nc = dy; len = dx; }
    const int64_t npix = nr * nc;
    if (npix == 0) return IVX_OK;
    IVX_REQUIRE(len > 0, IVX_EINVAL, "mip: zero-size array to reduction operation");
    void *part;
    int rc;
    if (axis == 2) {
        if ((rc = ivx::ws_get_s(ivx::WS_AUX3, st, (size_t)npix * 8, &part))) return rc;
        hipLaunchKernelGGL((k_reduce_rows<T, OP, VEC>), dim3((unsigned)ivx::cdiv(npix, 4)), dim3(256), 0, st, vol, npix,
                           len, (long long *)part);
        IVX_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_combine<T, OP, long long>), dim3((unsigned)ivx::cdiv(npix, 256)), dim3(256), 0, st,
                           (const long long *)part, npix, 1, len, out);
        IVX_LAUNCH_CHECK();
        return IVX_OK;
    }
    const int64_t sr = axis == 0 ? dx : dy * dx;
    const int64_t sl = axis == 0 ? dy * dx : dx;
    const int64_t ncg = ivx::cdiv(nc, VEC);
    const int64_t nthreads = nr * ncg;
    const int64_t nblk = ivx::cdiv(nthreads, 256);
    // split the ray so that >= ~2048 workgroups exist (segments >= 32 samples, <= 32767 so int32 partial sums hold)
    int64_t split = ivx::cdiv(2048, nblk);
    if (split > len / 32) split = len / 32;
    if (split < ivx::cdiv(len, 32767)) split = ivx::cdiv(len, 32767);
    if (split < 1) split = 1;
    const int64_t seg = ivx::cdiv(len, split);
    split = ivx::cdiv(len, seg);
    if ((rc = ivx::ws_get_s(ivx::WS_AUX3, st, (size_t)npix * sizeof(part_t) * split, &part))) return rc;
    hipLaunchKernelGGL((k_reduce_strided<T, OP, VEC>), dim3((unsigned)nblk, (unsigned)split), dim3(256), 0, st, vol, nr,
                       nc, len, sr, sl, seg, (part_t *)part);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_combine<T, OP, part_t>), dim3((unsigned)ivx::cdiv(npix, 256)), dim3(256), 0, st,
                       (const part_t *)part, npix, (int)split, len, out);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

template <typename T>
static int run_reduce_op(const void *vol, int64_t dz, int64_t dy, int64_t dx, int axis, int op, void *out,
                         hipStream_t st) {
    switch (op) {
    case IVX_MIP_MAX: return run_reduce<T, IVX_MIP_MAX>(vol, dz, dy, dx, axis, out, st);
    case IVX_MIP_MIN: return run_reduce<T, IVX_MIP_MIN>(vol, dz, dy, dx, axis, out, st);
    case IVX_MIP_MEAN: return run_reduce<T, IVX_MIP_MEAN>(vol, dz, dy, dx, axis, out, st);
    case IVX_MIP_SUM: return run_reduce<T, IVX_MIP_SUM>(vol, dz, dy, dx, axis, out, st);
    }
    ivx::set_error("mip: unknown op %d", op);
    return IVX_EINVAL;
}

} // namespace

extern "C" int ivx_dev_mip_reduce(int dtype, const void *vol, int64_t dz, int64_t dy, int64_t dx, int axis, int op,
                                  void *out, void *stream) {
    IVX_REQUIRE(axis >= 0 && axis <= 2, IVX_EINVAL, "mip: axis %d", axis);
    IVX_REQUIRE(dz >= 0 && dy >= 0 && dx >= 0, IVX_EINVAL, "mip: negative shape");
    hipStream_t st = ivx::S(stream);
    switch (dtype) {
    case IVX_I16: return run_reduce_op<int16_t>(vol, dz, dy, dx, axis, op, out, st);
    case IVX_U8: return run_reduce_op<uint8_t>(vol, dz, dy, dx, axis, op, out, st);
    case IVX_U16: return run_reduce_op<uint16_t>(vol, dz, dy, dx, axis, op, out, st);
    }
    ivx::set_error("mip: unsupported dtype %d (integer images only)", dtype);
    return IVX_EINVAL;
}

extern "C" int ivx_mip_reduce(int dtype, const void *vol, const int64_t shape[3], const int64_t strides[3], int axis,
                              int op, void *out, const int64_t out_strides[2]) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(axis >= 0 && axis <= 2, IVX_EINVAL, "mip: axis %d", axis);
    const size_t isz = dtype_size(dtype);
    IVX_REQUIRE(isz == 1 || isz == 2, IVX_EINVAL, "mip: unsupported dtype %d", dtype);
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    int64_t osh[2];
    if (axis == 0) { osh[0] = shape[1]; osh[1] = shape[2]; }
    else if (axis == 1) { osh[0] = shape[0]; osh[1] = shape[2]; }
    else { osh[0] = shape[0]; osh[1] = shape[1]; }
    const size_t osz = (op == IVX_MIP_MEAN || op == IVX_MIP_SUM) ? 8 : isz;
    void *d_in, *d_out;
    int rc;
    if ((rc = ws_get(WS_IN, n * isz, &d_in))) return rc;
    if ((rc = ws_get(WS_OUT, (size_t)osh[0] * osh[1] * osz, &d_out))) return rc;
    if ((rc = upload_strided(d_in, vol, shape, strides, isz, WS_IN))) return rc;
    if ((rc = ivx_dev_mip_reduce(dtype, d_in, shape[0], shape[1], shape[2], axis, op, d_out, nullptr))) return rc;
    IVX_HIP(hipDeviceSynchronize());
    return download_strided2(out, osh, out_strides, d_out, osz, WS_OUT);
}

// ---- viewport: nearest-sample magnification of a projection ------------------------------------------------------
// BASELINE configs[4] asks for a 2048 x 2048 viewport over a 512^3 volume: the reference's ray caster samples 4 x 4
// rays per voxel column (SetImageSampleDistance(0.25), invesalius/data/volume.py:678); for axis-aligned rays every one of
// them sees the same samples, so the viewport image IS the projection with each pixel repeated f x f times.
namespace {
__global__ __launch_bounds__(256) void k_replicate_i16(const int16_t *__restrict__ src, int64_t h, int64_t w, int f,
                                                       int16_t *__restrict__ dst) {
    const int64_t W = w * f, H = h * f;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; // 8 output pixels = one 16-byte store
    if (i >= W * H) return;
    const int64_t y = i / W, x = i - y * W;
    const int16_t *row = src + (y / f) * w;
    short v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = x + q < W ? row[(x + q) / f] : (short)0;
    if (x + 8 <= W && (W & 7) == 0) {
        *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(v);
    } else {
        for (int q = 0; q < 8 && i + q < W * H; q++) { // ragged width: pixel by pixel, rows may straddle the chunk
            const int64_t j = i + q, yy = j / W, xx = j - yy * W;
            dst[j] = src[(yy / f) * w + xx / f];
        }
    }
}
} // namespace

extern "C" int ivx_dev_replicate_i16(const int16_t *src, int64_t h, int64_t w, int factor, int16_t *dst, void *stream) {
    IVX_REQUIRE(src && dst && h > 0 && w > 0 && factor >= 1, IVX_EINVAL, "replicate: bad arguments");
    const int64_t n = h * factor * w * factor;
    hipLaunchKernelGGL(k_replicate_i16, dim3((unsigned)ivx::cdiv(ivx::cdiv(n, 8), 256)), dim3(256), 0, ivx::S(stream), src, h, w,
                       factor, dst);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
