// k_area.hip -- convolve_non_zero and the mask-area measurement built on it.
//
// Reference semantics:
//   convolve_non_zero   invesalius_rs/src/transforms_py.rs:51-93  out = 0 where volume == 0, else the CORRELATION of the
//                       float64 volume with the float64 kernel (no flip), samples outside the volume = cval, summed in
//                       (k, j, i) order -- restated term by term, so every output value is bit-identical
//   Slice.calc_image_area  invesalius/data/slice_.py:2296-2322    bin = mask > 127; area = convolve_non_zero(bin * 1.0,
//                       K(spacing), 1).sum(), K = the 7-point "exposed faces" kernel
// MI355X: streaming; the generic form is 8 B read (+ taps through L1/L2) and 8 B written per voxel; the area form reads
// the uint8 mask directly (1 B/voxel), never materialises the float64 volume (1 GiB at 512^3) and reduces in double
// with a fixed-shape tree (reproducible; numpy's pairwise sum differs from it by rounding only).
#include "ivx_internal.h"

namespace {

__global__ __launch_bounds__(256) void k_convolve_non_zero(const double *__restrict__ vol, int64_t sz, int64_t sy, int64_t sx,
                                                           const double *__restrict__ ker, int skz, int sky, int skx,
                                                           double cval, double *__restrict__ out) {
    const int64_t n = sz * sy * sx;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        double r = 0.0;
        if (vol[idx] != 0.0) {
            const int64_t x = idx % sx, t = idx / sx, y = t % sy, z = t / sy;
            double sum = 0.0;
            for (int k = 0; k < skz; k++) {
                const int64_t kz = z - skz / 2 + k;
                for (int j = 0; j < sky; j++) {
                    const int64_t ky = y - sky / 2 + j;
                    for (int i = 0; i < skx; i++) {
                        const int64_t kx = x - skx / 2 + i;
                        const bool in = kz >= 0 && kz < sz && ky >= 0 && ky < sy && kx >= 0 && kx < sx;
                        const double v = in ? vol[(kz * sy + ky) * sx + kx] : cval;
                        sum += v * ker[(k * sky + j) * skx + i];
                    }
                }
            }
            r = sum;
        }
        out[idx] = r;
    }
}

// per-voxel value of convolve_non_zero(bin * 1.0, K, 1) for the 3x3x3 kernel K, straight from the uint8 mask
__device__ __forceinline__ double area_term(const uint8_t *__restrict__ m, int64_t z, int64_t y, int64_t x, int64_t sz,
                                            int64_t sy, int64_t sx, const double *__restrict__ K) {
    // Only the seven axis taps of this kernel are non-zero.  The other twenty contribute v * (+0.0) = +0.0 for v in {0, 1},
    // which never changes a float64 sum, so they are skipped; the seven are added in the reference's (k, j, i) order.
    const int8_t tap[7][3] = {{0, 1, 1}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}, {1, 1, 2}, {1, 2, 1}, {2, 1, 1}};
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const int k = tap[q][0], j = tap[q][1], i = tap[q][2];
        const int64_t kz = z - 1 + k, ky = y - 1 + j, kx = x - 1 + i;
        const bool in = kz >= 0 && kz < sz && ky >= 0 && ky < sy && kx >= 0 && kx < sx;
        const double v = in ? (m[(kz * sy + ky) * sx + kx] > 127 ? 1.0 : 0.0) : 1.0; // cval = 1
        sum += v * K[(k * 3 + j) * 3 + i];
    }
    return sum;
}

constexpr int AREA_VPL = 8;
__global__ __launch_bounds__(256) void k_mask_area(const uint8_t *__restrict__ m, int64_t sz, int64_t sy, int64_t sx,
                                                   const double *__restrict__ K27, double *__restrict__ partial) {
    __shared__ double s_k[27];
    __shared__ double s_part[4];
    if (threadIdx.x < 27) s_k[threadIdx.x] = K27[threadIdx.x];
    __syncthreads();
    const int64_t n = sz * sy * sx;
    const int64_t base = (int64_t)blockIdx.x * 256 * AREA_VPL;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < AREA_VPL; q++) {
        const int64_t idx = base + (int64_t)q * 256 + threadIdx.x;
        if (idx < n && m[idx] > 127) {
            const int64_t x = idx % sx, t = idx / sx;
            acc += area_term(m, t / sy, t % sy, x, sz, sy, sx, s_k);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}
__global__ __launch_bounds__(256) void k_sum_partials(const double *__restrict__ partial, int64_t nb, double *__restrict__ out) {
    __shared__ double s_part[4];
    double acc = 0.0;
    for (int64_t b = threadIdx.x; b < nb; b += 256) acc += partial[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

} // namespace

extern "C" int ivx_dev_convolve_non_zero(const double *volume, int64_t sz, int64_t sy, int64_t sx, const double *kernel,
                                         int64_t skz, int64_t sky, int64_t skx, int cval, double *out, void *stream) {
    IVX_REQUIRE(sz >= 0 && sy >= 0 && sx >= 0 && skz >= 0 && sky >= 0 && skx >= 0, IVX_EINVAL, "convolve_non_zero: negative shape");
    IVX_REQUIRE(skz < 4096 && sky < 4096 && skx < 4096, IVX_EINVAL, "convolve_non_zero: kernel too large");
    IVX_REQUIRE(cval >= -32768 && cval <= 32767, IVX_ERANGE, "convolve_non_zero: cval does not fit an int16");
    const int64_t n = sz * sy * sx;
    if (n == 0) return IVX_OK;
    const int64_t blocks = ivx::cdiv(n, 256);
    hipLaunchKernelGGL(k_convolve_non_zero, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, ivx::S(stream),
                       volume, sz, sy, sx, kernel, (int)skz, (int)sky, (int)skx, (double)cval, out);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

extern "C" int ivx_dev_mask_area_u8(const uint8_t *mask, int64_t sz, int64_t sy, int64_t sx, const double *kernel27_dev,
                                    double *scratch, double *area_dev, void *stream) {
    IVX_REQUIRE(sz >= 0 && sy >= 0 && sx >= 0, IVX_EINVAL, "mask_area: negative shape");
    hipStream_t st = ivx::S(stream);
    const int64_t n = sz * sy * sx;
    if (n == 0) {
        IVX_HIP(hipMemsetAsync(area_dev, 0, 8, st));
        return IVX_OK;
    }
    const int64_t nb = ivx::cdiv(n, (int64_t)256 * AREA_VPL);
    hipLaunchKernelGGL(k_mask_area, dim3((unsigned)nb), dim3(256), 0, st, mask, sz, sy, sx, kernel27_dev, scratch);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, (const double *)scratch, nb, area_dev);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}
extern "C" int ivx_mask_area_scratch_bytes(int64_t sz, int64_t sy, int64_t sx, size_t *nbytes) {
    *nbytes = (size_t)ivx::cdiv(sz * sy * sx, (int64_t)256 * AREA_VPL) * 8 + 64;
    return IVX_OK;
}

// ---- host forms ------------------------------------------------------------------------------------------------------
extern "C" int ivx_convolve_non_zero(const double *volume, const int64_t shape[3], const int64_t vst[3], const double *kernel,
                                     const int64_t kshape[3], int cval, double *out) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0, IVX_EINVAL, "convolve_non_zero: negative shape");
    const size_t n = (size_t)shape[0] * shape[1] * shape[2], nk = (size_t)kshape[0] * kshape[1] * kshape[2];
    if (n == 0) return IVX_OK;
    void *d_v, *d_k, *d_o;
    int rc;
    if ((rc = ws_get(WS_IN, n * 8, &d_v))) return rc;
    if ((rc = ws_get(WS_OUT, n * 8, &d_o))) return rc;
    if ((rc = ws_get(WS_SMALL, nk * 8 + 16, &d_k))) return rc;
    if ((rc = upload_strided(d_v, volume, shape, vst, 8, WS_IN))) return rc;
    if (nk) IVX_HIP(hipMemcpy(d_k, kernel, nk * 8, hipMemcpyHostToDevice));
    if ((rc = ivx_dev_convolve_non_zero((const double *)d_v, shape[0], shape[1], shape[2], (const double *)d_k, kshape[0],
                                        kshape[1], kshape[2], cval, (double *)d_o, nullptr)))
        return rc;
    IVX_HIP(hipMemcpy(out, d_o, n * 8, hipMemcpyDeviceToHost));
    return IVX_OK;
}

// mask: the uint8 interior view (strided); area = sum over voxels > 127 of the exposed-face kernel (slice_.py:2296-2322)
extern "C" int ivx_mask_area(const uint8_t *mask, const int64_t shape[3], const int64_t mst[3], const double spacing_xyz[3],
                             double *area) {
    ivx::HostCallGuard host_guard__;
    using namespace ivx;
    IVX_REQUIRE(shape[0] >= 0 && shape[1] >= 0 && shape[2] >= 0, IVX_EINVAL, "mask_area: negative shape");
    *area = 0.0;
    const size_t n = (size_t)shape[0] * shape[1] * shape[2];
    if (n == 0) return IVX_OK;
    const double sx = spacing_xyz[0], sy = spacing_xyz[1], sz = spacing_xyz[2];
    double K[27];
    for (int q = 0; q < 27; q++) K[q] = 0.0;
    K[13] = 2 * sx * sy + 2 * sx * sz + 2 * sy * sz; // kernel[1,1,1]
    K[4] = -(sx * sy);                               // kernel[0,1,1]
    K[22] = -(sx * sy);                              // kernel[2,1,1]
    K[10] = -(sx * sz);                              // kernel[1,0,1]
    K[16] = -(sx * sz);                              // kernel[1,2,1]
    K[12] = -(sy * sz);                              // kernel[1,1,0]
    K[14] = -(sy * sz);                              // kernel[1,1,2]
    size_t sb;
    ivx_mask_area_scratch_bytes(shape[0], shape[1], shape[2], &sb);
    void *d_m, *d_s, *d_k;
    int rc;
    if ((rc = ws_get(WS_IN, n, &d_m))) return rc;
    if ((rc = ws_get(WS_AUX0, sb, &d_s))) return rc;
    if ((rc = ws_get(WS_SMALL, 27 * 8 + 64, &d_k))) return rc;
    if ((rc = upload_strided(d_m, mask, shape, mst, 1, WS_IN))) return rc;
    IVX_HIP(hipMemcpy(d_k, K, sizeof(K), hipMemcpyHostToDevice));
    double *d_area = (double *)((char *)d_k + 27 * 8 + 8);
    if ((rc = ivx_dev_mask_area_u8((const uint8_t *)d_m, shape[0], shape[1], shape[2], (const double *)d_k, (double *)d_s, d_area,
                                   nullptr)))
        return rc;
    IVX_HIP(hipMemcpy(area, d_area, 8, hipMemcpyDeviceToHost));
    return IVX_OK;
}
