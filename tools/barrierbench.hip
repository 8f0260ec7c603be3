// dev micro-benchmark: cost of an in-kernel grid barrier (monotonic counter, agent-scope atomics) on MI355X
// build: hipcc --offload-arch=gfx950 -O3 -o tools/barrierbench tools/barrierbench.hip ; run: tools/barrierbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned long long* bar, unsigned long long target, int variant) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (variant == 0) {
            __threadfence();
            __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        } else if (variant == 1) {
            __atomic_thread_fence(__ATOMIC_RELEASE);      // agent scope by default for the device
            __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        } else {
            // two levels: 16 blocks per group counter (bar[1 + g*16] on its own 128-byte line), the last of a group bumps the root
            const int g = blockIdx.x >> 4, ng = (gridDim.x + 15) >> 4;
            const unsigned long long round = target / gridDim.x;          // 1, 2, 3, ...
            const int gsize = min(16, (int)gridDim.x - g * 16);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            const unsigned long long prev = __hip_atomic_fetch_add(bar + 16 * (1 + g), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1 == round * gsize) __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * ng) {}
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_bar(unsigned long long* bar, int n, int* sink, int work, int variant) {
    extern __shared__ unsigned char lds[];
    unsigned long long t = 0;
    int acc = 0;
    for (int r = 0; r < n; ++r) {
        if (work) { sink[blockIdx.x * 256 + threadIdx.x] = r; acc += sink[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x]; }
        t += gridDim.x;
        grid_barrier(bar, t, variant);
    }
    if (acc == -1) sink[0] = acc;
}

int main() {
    unsigned long long* bar; int* sink;
    CHK(hipMalloc(&bar, 8 * 16 * 64)); CHK(hipMalloc(&sink, 1024 * 256 * 4));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bar), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int blocks : {64, 128, 256}) for (int variant : {0, 1, 2}) for (int work : {1}) { const int lds = 0;
        const int n = 2000;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipMemset(bar, 0, 8 * 16 * 64));
            void* args[] = {&bar, (void*)&n, &sink, (void*)&work, (void*)&variant};
            CHK(hipEventRecord(a));
            CHK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_bar), dim3(blocks), dim3(256), args, lds, nullptr));
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("blocks %4d variant %d work %d: %.2f us per barrier\n", blocks, variant, work, best * 1000.f / n);
    }
    return 0;
}
