// scan_u32.h -- exclusive scan of a uint32 array in place, three launches (4096 elements per workgroup).
// Included by the translation units that need it (kernels live in each unit's anonymous namespace).
//   scan_u32_exclusive(data, n, bsum, total, stream): bsum = scratch of cdiv(n, 4096) + 1 words, *total (device) = sum
#pragma once
#include "ivx_internal.h"

namespace {
constexpr int MSCAN = 16;
__global__ __launch_bounds__(256) void k_mscan_block(uint32_t *__restrict__ data, int64_t n, uint32_t *__restrict__ bsum) {
    __shared__ uint32_t s_wave[4];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * MSCAN;
    uint32_t v[MSCAN], sum = 0;
#pragma unroll
    for (int q = 0; q < MSCAN; q++) {
        v[q] = base + q < n ? data[base + q] : 0u;
        sum += v[q];
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
    for (int q = 0; q < MSCAN; q++) {
        if (base + q < n) data[base + q] = off;
        off += v[q];
    }
    if (threadIdx.x == 255) bsum[blockIdx.x] = off;
}
__global__ __launch_bounds__(1024) void k_mscan_sums(uint32_t *__restrict__ bsum, int64_t nb, uint32_t *__restrict__ total) {
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
__global__ __launch_bounds__(256) void k_mscan_add(uint32_t *__restrict__ data, int64_t n, const uint32_t *__restrict__ bsum) {
    const uint32_t add = bsum[blockIdx.x];
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * MSCAN;
#pragma unroll
    for (int q = 0; q < MSCAN; q++)
        if (base + q < n) data[base + q] += add;
}


static inline int64_t scan_u32_blocks(int64_t n) { return ivx::cdiv(n, 256 * MSCAN); }
static inline int scan_u32_exclusive(uint32_t *data, int64_t n, uint32_t *bsum, uint32_t *total, hipStream_t st) {
    const int64_t nsb = scan_u32_blocks(n);
    if (n <= 0) {
        IVX_HIP(hipMemsetAsync(total, 0, 4, st));
        return IVX_OK;
    }
    hipLaunchKernelGGL(k_mscan_block, dim3((unsigned)nsb), dim3(256), 0, st, data, n, bsum);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mscan_sums, dim3(1), dim3(1024), 0, st, bsum, nsb, total);
    IVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mscan_add, dim3((unsigned)nsb), dim3(256), 0, st, data, n, bsum);
    IVX_LAUNCH_CHECK();
    return IVX_OK;
}

} // namespace
