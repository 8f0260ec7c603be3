// tools/gatherbench3.hip — dev micro-benchmark (not product code): the "every wave both queues" experiment of gatherbench2.hip
// (E5) with the three sizes on the command line, so that the stream / gather floor can be measured for any PCSR shape:
//   gatherbench3 <log2 slots> <gathers> <x entries>
//     slots    12-byte PMA slots streamed once (int32 key + f64 value)
//     gathers  8-byte reads from x at uniformly random indices (one per stored cell)
//     x        entries of the gathered table (8 B each; any size, not only powers of two)
// config 3 (10 M nnz, 2^24 slots, 1 M columns):      gatherbench3 24 10000000 1000000
// a config-4 shard (12.5 M nnz, 2^25 slots, 1.25 M):  gatherbench3 25 12500000 1250000
// Build: hipcc --offload-arch=gfx950 -O3 -o gatherbench3 gatherbench3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int W = 8;            // words (of 64 slots) per span

__device__ __forceinline__ uint64_t mixh(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// every wave takes stream spans (512 slots each) and gather spans (512 gathers each) in turn, grid-strided
__global__ __launch_bounds__(256) void k_both(const int32_t* __restrict__ kp, const double* __restrict__ vp, const double* __restrict__ x,
                                              uint32_t nx, int nstream, int ngather, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    double acc = 0.0;
    const int first = blockIdx.x * 4 + wv, step = gridDim.x * 4;
    for (int s = first; s < nstream; s += step) {
        int32_t k[W]; double v[W];
        const int64_t b = (int64_t)s * (W * 64) + lane;
#pragma unroll
        for (int j = 0; j < W; ++j) { k[j] = __builtin_nontemporal_load(kp + b + j * 64); v[j] = __builtin_nontemporal_load(vp + b + j * 64); }
#pragma unroll
        for (int j = 0; j < W; ++j) acc += v[j] + (double)k[j];
    }
    for (int s = first; s < ngather; s += step) {
        double t[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const uint32_t h = (uint32_t)mixh(((uint64_t)s * 64 + lane) * 1315423911ull + (uint64_t)j * 0x9E3779B97F4A7C15ull);
            t[j] = x[(uint32_t)(((uint64_t)h * nx) >> 32)];
        }
#pragma unroll
        for (int j = 0; j < W; ++j) acc += t[j];
    }
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// the key-driven form (gatherbench2's E1): one wave per 512 slots, a gather per stored cell (key >= 0), one sum per wave.
// PRED: gap lanes issue no gather (exec-masked) instead of reading x[0]
template <bool PRED>
__global__ __launch_bounds__(256) void k_keyed(const int32_t* __restrict__ kp, const double* __restrict__ vp, const double* __restrict__ x,
                                               int nstream, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    // XCD g streams the g-th eighth of the spans (the product's mapping)
    const int per = nstream / 8;
    const int s = (blockIdx.x & 7) * per + (blockIdx.x >> 3) * 4 + wv;
    if ((int)(blockIdx.x >> 3) * 4 + wv >= per) return;
    int32_t k[W]; double v[W], xv[W];
    const int64_t b = (int64_t)s * (W * 64) + lane;
#pragma unroll
    for (int j = 0; j < W; ++j) { k[j] = __builtin_nontemporal_load(kp + b + j * 64); v[j] = __builtin_nontemporal_load(vp + b + j * 64); }
#pragma unroll
    for (int j = 0; j < W; ++j) {
        if (PRED) { xv[j] = 0.0; if (k[j] >= 0) xv[j] = x[k[j]]; }
        else xv[j] = x[k[j] >= 0 ? k[j] : 0];
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < W; ++j) acc += k[j] >= 0 ? v[j] * xv[j] : 0.0;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[s] = acc;
}

template <typename F> static float timeit(F f, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: gatherbench3 <log2 slots> <gathers> <x entries>\n"); return 2; }
    const int lg = atoi(argv[1]);
    const int64_t G = atoll(argv[2]);
    const int64_t NX = atoll(argv[3]);
    if (lg < 16 || lg > 28 || G < 512 || G > ((int64_t)1 << 30) || NX < 1 || NX > ((int64_t)1 << 28)) { printf("sizes out of range\n"); return 2; }
    const int64_t S = (int64_t)1 << lg;
    const int nstream = (int)(S / (W * 64)), ngather = (int)(G / (W * 64));
    int32_t* keys; double *vals, *x, *out;
    CK(hipMalloc(&keys, S * 4)); CK(hipMalloc(&vals, S * 8)); CK(hipMalloc(&x, (size_t)NX * 8)); CK(hipMalloc(&out, (size_t)8192 * 256 * 8));
    CK(hipMemset(keys, 0, S * 4)); CK(hipMemset(vals, 0, S * 8)); CK(hipMemset(x, 0, (size_t)NX * 8));
    printf("slots 2^%d (stream %.1f MB), %lld gathers (%d spans), x %.2f MB\n", lg, S * 12 / 1e6, (long long)G, ngather, NX * 8 / 1e6);
    for (int grid : {1024, 2048, 4096, 8192}) {
        auto run = [&](int ns, int ng) { return timeit([&] { hipLaunchKernelGGL(k_both, dim3(grid), dim3(256), 0, 0, keys, vals, x, (uint32_t)NX, ns, ng, out); }); };
        printf("  grid %d: stream only %.1f us | gather only %.1f | both %.1f\n", grid, run(nstream, 0), run(0, ngather), run(nstream, ngather));
    }
    {
        // real keys: every slot holds a cell with probability G / S, key uniform in [0, NX)
        std::vector<int32_t> hk((size_t)S);
        uint64_t st = 777; int64_t cells = 0;
        const uint64_t thr = (uint64_t)((double)G / (double)S * 4294967296.0);
        for (int64_t i = 0; i < S; ++i) {
            uint64_t z = (st += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            if ((z & 0xffffffffull) < thr) { hk[(size_t)i] = (int32_t)(((z >> 32) * (uint64_t)NX) >> 32); ++cells; } else hk[(size_t)i] = -1;
        }
        CK(hipMemcpy(keys, hk.data(), (size_t)S * 4, hipMemcpyHostToDevice));
        const int grid = (nstream / 8 + 3) / 4 * 8;
        printf("key-driven, one wave per 512 slots (%lld cells, grid %d): gap lanes read x[0] %.1f us | gap lanes masked off %.1f us\n", (long long)cells, grid,
               timeit([&] { hipLaunchKernelGGL(k_keyed<false>, dim3(grid), dim3(256), 0, 0, keys, vals, x, nstream, out); }),
               timeit([&] { hipLaunchKernelGGL(k_keyed<true>, dim3(grid), dim3(256), 0, 0, keys, vals, x, nstream, out); }));
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
