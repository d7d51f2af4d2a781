// clock_probe.hip -- what clock do latency-bound kernels run at?  One workgroup times (a) a dependent ALU chain and (b) a
// dependent pointer chase through 256 MiB, alone on the chip and next to a background kernel that keeps every CU busy
// (ALU-only spinner / HBM streamer).  hipcc --offload-arch=gfx950 -O2 tools/micro/clock_probe.hip -o tools/micro/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_alu(unsigned long long *out, int iters) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned x = threadIdx.x + 1;
    for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = x; }
}

template <int MODE> __global__ void k_chase(const unsigned *next, unsigned long long *out, int steps) {
    const unsigned long long w0 = wall_clock64();
    unsigned p = threadIdx.x * 977u;
    for (int i = 0; i < steps; i++) {
        if (MODE == 0) p = next[p];
        else p = __hip_atomic_load(&next[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[2] = p; }
}

__global__ void k_spin(volatile int *stop, unsigned *sink) { // ALU-only background: every CU busy, no memory traffic
    unsigned x = threadIdx.x;
    while (!*stop) {
        for (int i = 0; i < 4096; i++) x = x * 1664525u + 1013904223u;
    }
    if (x == 12345u) sink[0] = x;
}

__global__ void k_stream(volatile int *stop, const uint4 *src, uint4 *dst, size_t n) { // HBM streaming background
    uint4 acc = {0, 0, 0, 0};
    while (!*stop) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const uint4 v = src[i];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    dst[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const size_t N = 64u << 20; // 256 MiB of uint32
    std::vector<unsigned> perm(N);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937_64 rng(1);
    for (size_t i = N - 1; i > 0; i--) { size_t j = rng() % (i + 1); std::swap(perm[i], perm[j]); } // one random cycle-ish permutation
    unsigned *d_next; unsigned long long *d_out; int *d_stop; unsigned *d_sink; uint4 *d_src, *d_dst;
    CK(hipMalloc(&d_next, N * 4)); CK(hipMemcpy(d_next, perm.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 64)); CK(hipHostMalloc(&d_stop, 4)); CK(hipMalloc(&d_sink, 64));
    const size_t SN = 64u << 20; CK(hipMalloc(&d_src, SN * 16)); CK(hipMalloc(&d_dst, 1 << 24)); CK(hipMemset(d_src, 1, SN * 16));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long h[3];
    for (int bg = 0; bg < 4; bg++) { // 0 alone, 1 spinner on 1 wave per CU, 2 spinner on 4 waves/CU, 3 streamer
        *d_stop = 0;
        if (bg == 1) hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, s2, d_stop, d_sink);
        if (bg == 2) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s2, d_stop, d_sink);
        if (bg == 3) hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, s2, d_stop, d_src, d_dst, SN);
        const char *nm[] = {"alone", "spinner 1 wave/CU", "spinner 4 waves/CU", "HBM streamer"};
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_alu, dim3(1), dim3(64), 0, s1, d_out, 1 << 20);
            CK(hipMemcpyAsync(h, d_out, 24, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1));
            const double us = h[0] * 0.01;
            printf("%-20s alu chain: %.1f us for 2^20 mul-add -> %.2f ns/iter, clock64 %.0f ticks/us\n", nm[bg], us, us * 1e3 / (1 << 20), h[1] / us);
            fflush(stdout);
        }
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 0) hipLaunchKernelGGL(k_chase<0>, dim3(1), dim3(64), 0, s1, d_next, d_out, 2000);
            else hipLaunchKernelGGL(k_chase<1>, dim3(1), dim3(64), 0, s1, d_next, d_out, 2000);
            CK(hipMemcpyAsync(h, d_out, 24, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1));
            printf("%-20s pointer chase (%s loads, 64 lanes): %.0f ns per dependent load\n", nm[bg], mode ? "agent-scope" : "plain", h[0] * 10.0 / 2000);
        }
        *d_stop = 1;
        CK(hipStreamSynchronize(s2));
        fflush(stdout);
    }
    // clocks as the SMI would report them are not readable without privileges; the ALU chain is the measurement
    return 0;
}
