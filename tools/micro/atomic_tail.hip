// micro-benchmark: what do ~10^3 atomics on ONE counter at the end of a kernel cost (the flood rounds' list appends)?
// hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_tail.hip -o gpurun_out/atomic_tail && gpurun_out/atomic_tail
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned int *cnt, int nsub, int stride_words, int work, unsigned int *sink, int per_wg) {
    unsigned int x = threadIdx.x;
    for (int i = 0; i < work; i++) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) sink[0] = x;
    __syncthreads();
    if ((int)threadIdx.x < per_wg && nsub > 0) {
        const unsigned int s = (blockIdx.x * per_wg + threadIdx.x) % nsub;
        const unsigned int old = atomicAdd(&cnt[s * stride_words], 1u);
        sink[1 + old % 1024] = blockIdx.x; // (the append itself)
    }
}
int main() {
    unsigned int *cnt, *sink;
    hipMalloc(&cnt, 1 << 20);
    hipMalloc(&sink, 1 << 16);
    hipMemset(cnt, 0, 1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grids[] = {256, 1024, 1536};
    for (int gi = 0; gi < 3; gi++)
        for (int per_wg = 1; per_wg <= 4; per_wg *= 4)
            for (int nsub : {0, 1, 8, 64}) {
                for (int it = 0; it < 20; it++) hipLaunchKernelGGL(k, dim3(grids[gi]), dim3(256), 0, 0, cnt, nsub, 32, 2000, sink, per_wg);
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                const int reps = 200;
                for (int it = 0; it < reps; it++) hipLaunchKernelGGL(k, dim3(grids[gi]), dim3(256), 0, 0, cnt, nsub, 32, 2000, sink, per_wg);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                printf("grid %4d appends/wg %d counters %2d : %.2f us per launch\n", grids[gi], per_wg, nsub, ms * 1000.f / reps);
            }
    return 0;
}
