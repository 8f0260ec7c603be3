// tools/launchbench.hip — dev micro-benchmark: what does launching N workgroups of B threads cost on gfx950 when the kernel
// does nothing (dispatch ramp), and with 24 KB of LDS per workgroup (the occupancy limit of k_move2)?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS> __global__ void k_empty(int* out) {
    __shared__ int s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (LDS > 0) { s[threadIdx.x] = threadIdx.x; __syncthreads(); if (out && s[(threadIdx.x + 1) % blockDim.x] == -1) out[0] = 1; }
}
template <typename F> static float timeit(F f, int reps = 50) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}
int main() {
    int* out; (void)hipMalloc(&out, 4);
    for (int B : {64, 256, 512, 1024}) {
        printf("block %4d:", B);
        for (int N : {1, 64, 256, 512, 1024, 2048, 4096, 8192}) {
            float t0 = timeit([&] { hipLaunchKernelGGL(k_empty<0>, dim3(N), dim3(B), 0, 0, out); });
            float t1 = timeit([&] { hipLaunchKernelGGL(k_empty<24576>, dim3(N), dim3(B), 0, 0, out); });
            printf("  N=%d %.1f/%.1f", N, t0, t1);
        }
        printf("  us (no LDS / 24 KB LDS)\n");
    }
    return 0;
}
