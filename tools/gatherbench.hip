// tools/gatherbench.hip — dev micro-benchmark (not product code): what is the hardware ceiling of the two
// ingredients of the PCSR SpMV on gfx950?
//   (A) random 8-byte gathers from a table of T bytes (indices hashed in registers: no other memory traffic),
//       with plain / nt / sc1 loads and 4..16 gathers in flight per lane;
//   (B) a coalesced stream of 8-byte keys + 8-byte values at 8 B/lane vs 16 B/lane;
//   (C) both together: stream (key,val), gather x[key], sum.
// Build: hipcc --offload-arch=gfx950 -O3 -o gatherbench gatherbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int MODE> __device__ __forceinline__ double ld(const double* p) {
    if (MODE == 0) return *p;
    if (MODE == 1) return __builtin_nontemporal_load(p);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (A) pure gathers: each lane does rounds*U gathers, U in flight
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_gather(const double* __restrict__ x, uint32_t mask, int rounds, double* __restrict__ out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        double t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t idx = (uint32_t)mix(gid * 1315423911ull + (uint64_t)(r * U + u) * 0x9E3779B97F4A7C15ull) & mask;
            t[u] = ld<MODE>(x + idx);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += t[u];
    }
    if (acc == 123.456) out[gid] = acc;
}

// (B) stream: W = 1 -> 8 B per lane per load, W = 2 -> 16 B per lane per load
template <int W, bool NT>
__global__ __launch_bounds__(256) void k_stream(const int64_t* __restrict__ keys, const double* __restrict__ vals, int64_t n, double* __restrict__ out) {
    // one workgroup per 2048-slot tile, 8 slots per thread
    const int64_t b0 = (int64_t)blockIdx.x * 2048;
    double acc = 0.0;
    if (W == 1) {
        int64_t k[8]; double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t s = b0 + j * 256 + threadIdx.x;
            k[j] = NT ? __builtin_nontemporal_load(keys + s) : keys[s];
            v[j] = NT ? __builtin_nontemporal_load(vals + s) : vals[s];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j] + (double)k[j];
    } else {
        typedef int64_t i2 __attribute__((ext_vector_type(2)));
        typedef double d2 __attribute__((ext_vector_type(2)));
        i2 k[4]; d2 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t s = b0 + j * 512 + threadIdx.x * 2;
            k[j] = NT ? __builtin_nontemporal_load((const i2*)(keys + s)) : *(const i2*)(keys + s);
            v[j] = NT ? __builtin_nontemporal_load((const d2*)(vals + s)) : *(const d2*)(vals + s);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y + (double)(k[j].x + k[j].y);
    }
    if (acc == 123.456) out[b0 + threadIdx.x] = acc;
}

// (C) stream + gather; occupied fraction ~0.656 emulated by key < 0 => gap
template <int MODE, int W>
__global__ __launch_bounds__(256) void k_both(const int64_t* __restrict__ keys, const double* __restrict__ vals, const double* __restrict__ x,
                                              double* __restrict__ out) {
    const int64_t b0 = (int64_t)blockIdx.x * 2048;
    double acc = 0.0;
    if (W == 1) {
        int64_t k[8]; double v[8], xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t s = b0 + j * 256 + threadIdx.x;
            k[j] = __builtin_nontemporal_load(keys + s);
            v[j] = __builtin_nontemporal_load(vals + s);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = k[j] >= 0 ? ld<MODE>(x + k[j]) : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j] * xv[j];
    } else {
        typedef int64_t i2 __attribute__((ext_vector_type(2)));
        typedef double d2 __attribute__((ext_vector_type(2)));
        i2 k[4]; d2 v[4]; double xa[4], xb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t s = b0 + j * 512 + threadIdx.x * 2;
            k[j] = __builtin_nontemporal_load((const i2*)(keys + s));
            v[j] = __builtin_nontemporal_load((const d2*)(vals + s));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xa[j] = k[j].x >= 0 ? ld<MODE>(x + k[j].x) : 0.0;
            xb[j] = k[j].y >= 0 ? ld<MODE>(x + k[j].y) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j].x * xa[j] + v[j].y * xb[j];
    }
    if (acc == 123.456) out[b0 + threadIdx.x] = acc;
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


// (D) propagation blocking: phase 1 bins (row, v*x) by row range with direct 16-B stores, phase 2 accumulates a bin in LDS
typedef int64_t Ent __attribute__((ext_vector_type(2)));   // (row, bits of the product)
template <int SLOTS_PER_THREAD, int LOGBW>
__global__ __launch_bounds__(256) void k_pb1(const int64_t* __restrict__ keys, const double* __restrict__ vals, int nb, int64_t bin_cap,
                                             unsigned* __restrict__ cursor, Ent* __restrict__ bins) {
    extern __shared__ unsigned sh[];           // cnt[nb], base[nb]
    unsigned* cnt = sh; unsigned* base = sh + nb;
    const int64_t b0 = (int64_t)blockIdx.x * (256 * SLOTS_PER_THREAD);
    for (int i = threadIdx.x; i < nb; i += 256) cnt[i] = 0;
    int64_t k[SLOTS_PER_THREAD]; double v[SLOTS_PER_THREAD]; unsigned rk[SLOTS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SLOTS_PER_THREAD; ++j) {
        const int64_t s = b0 + j * 256 + threadIdx.x;
        k[j] = __builtin_nontemporal_load(keys + s);
        v[j] = __builtin_nontemporal_load(vals + s);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SLOTS_PER_THREAD; ++j) if (k[j] >= 0) rk[j] = atomicAdd(&cnt[k[j] >> LOGBW], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256) { const unsigned c = cnt[i]; base[i] = c ? atomicAdd(&cursor[i], c) : 0u; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SLOTS_PER_THREAD; ++j) if (k[j] >= 0) {
        const int b = (int)(k[j] >> LOGBW);
        Ent e; e.x = k[j]; e.y = __double_as_longlong(v[j] * 1.5);
        bins[(int64_t)b * bin_cap + base[b] + rk[j]] = e;
    }
}
template <int LOGBW>
__global__ __launch_bounds__(1024) void k_pb2(const unsigned* __restrict__ cursor, const Ent* __restrict__ bins, int64_t bin_cap,
                                              double* __restrict__ y) {
    extern __shared__ double ys[];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < (1 << LOGBW); i += 1024) ys[i] = 0.0;
    __syncthreads();
    const unsigned n = cursor[b];
    const Ent* src = bins + (int64_t)b * bin_cap;
    unsigned i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        const Ent e0 = __builtin_nontemporal_load(src + i), e1 = __builtin_nontemporal_load(src + i + 1024), e2 = __builtin_nontemporal_load(src + i + 2048), e3 = __builtin_nontemporal_load(src + i + 3072);
        atomicAdd(&ys[e0.x & ((1 << LOGBW) - 1)], __longlong_as_double(e0.y)); atomicAdd(&ys[e1.x & ((1 << LOGBW) - 1)], __longlong_as_double(e1.y));
        atomicAdd(&ys[e2.x & ((1 << LOGBW) - 1)], __longlong_as_double(e2.y)); atomicAdd(&ys[e3.x & ((1 << LOGBW) - 1)], __longlong_as_double(e3.y));
    }
    for (; i < n; i += 1024) { const Ent e = src[i]; atomicAdd(&ys[e.x & ((1 << LOGBW) - 1)], __longlong_as_double(e.y)); }
    __syncthreads();
    for (int i2 = threadIdx.x; i2 < (1 << LOGBW); i2 += 1024) y[((int64_t)b << LOGBW) + i2] = ys[i2];
}
template <int SPT, int LOGBW> static void run_pb(const int64_t* keys, const double* vals, int64_t S, int64_t nrows, double* y, int64_t ng) {
    const int nb = (int)((nrows + (1 << LOGBW) - 1) >> LOGBW);
    const int64_t bin_cap = (int64_t)(ng / nb * 1.3) + 4096;
    unsigned* cursor; Ent* bins;
    hipMalloc(&cursor, nb * 4); hipMalloc(&bins, (size_t)nb * bin_cap * sizeof(Ent));
    const int chunks = (int)(S / (256 * SPT));
    float t1 = timeit([&] { hipMemsetAsync(cursor, 0, nb * 4, 0);
        hipLaunchKernelGGL((k_pb1<SPT, LOGBW>), dim3(chunks), dim3(256), nb * 8, 0, keys, vals, nb, bin_cap, cursor, bins); });
    float t2 = timeit([&] { hipLaunchKernelGGL((k_pb2<LOGBW>), dim3(nb), dim3(1024), (8 << LOGBW), 0, cursor, bins, bin_cap, y); });
    float t12 = timeit([&] { hipMemsetAsync(cursor, 0, nb * 4, 0);
        hipLaunchKernelGGL((k_pb1<SPT, LOGBW>), dim3(chunks), dim3(256), nb * 8, 0, keys, vals, nb, bin_cap, cursor, bins);
        hipLaunchKernelGGL((k_pb2<LOGBW>), dim3(nb), dim3(1024), (8 << LOGBW), 0, cursor, bins, bin_cap, y); });
    printf("(D) PB slots/thread %d, bin width %d rows (%d bins): phase1 %.1f us, phase2 %.1f us, both %.1f us\n", SPT, 1 << LOGBW, nb, t1, t2, t12);
    hipFree(cursor); hipFree(bins);
}


// (E) propagation blocking, phase 1 with an LDS counting sort by bin: coalesced runs in the bin stores
template <int SPT, int LOGBW, int THREADS>
__global__ __launch_bounds__(THREADS) void k_pb1s(const int64_t* __restrict__ keys, const double* __restrict__ vals, int nb, int64_t bin_cap,
                                                  unsigned* __restrict__ cursor, Ent* __restrict__ bins) {
    constexpr int T = SPT * THREADS;
    extern __shared__ unsigned char smem[];
    Ent* ent = (Ent*)smem;                                  // T entries
    unsigned short* ebin = (unsigned short*)(ent + T);      // T
    unsigned* cnt = (unsigned*)(ebin + T);                  // nb
    unsigned* scan = cnt + nb;                              // nb
    unsigned* gbase = scan + nb;                            // nb
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * T;
    for (int i = tid; i < nb; i += THREADS) cnt[i] = 0;
    int64_t k[SPT]; double v[SPT]; unsigned rk[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const int64_t s = b0 + j * THREADS + tid;
        k[j] = __builtin_nontemporal_load(keys + s);
        v[j] = __builtin_nontemporal_load(vals + s);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SPT; ++j) if (k[j] >= 0) rk[j] = atomicAdd(&cnt[k[j] >> LOGBW], 1u);
    __syncthreads();
    if (tid < 64) {                       // exclusive scan of cnt by one wave
        const int per = (nb + 63) / 64;
        unsigned loc = 0;
        for (int i = 0; i < per; ++i) { const int b = tid * per + i; if (b < nb) loc += cnt[b]; }
        unsigned inc = loc;
        for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(inc, d, 64); if (tid >= d) inc += t; }
        unsigned run = inc - loc;
        for (int i = 0; i < per; ++i) { const int b = tid * per + i; if (b < nb) { scan[b] = run; run += cnt[b]; } }
    }
    for (int i = tid; i < nb; i += THREADS) { const unsigned c = cnt[i]; gbase[i] = c ? atomicAdd(&cursor[i], c) : 0u; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SPT; ++j) if (k[j] >= 0) {
        const int b = (int)(k[j] >> LOGBW);
        const unsigned pos = scan[b] + rk[j];
        Ent e; e.x = k[j]; e.y = __double_as_longlong(v[j] * 1.5);
        ent[pos] = e; ebin[pos] = (unsigned short)b;
    }
    __syncthreads();
    const unsigned total = scan[nb - 1] + cnt[nb - 1];
    for (unsigned i = tid; i < total; i += THREADS) {
        const int b = ebin[i];
        bins[(int64_t)b * bin_cap + gbase[b] + (i - scan[b])] = ent[i];
    }
}
template <int SPT, int LOGBW, int THREADS> static void run_pbs(const int64_t* keys, const double* vals, int64_t S, int64_t nrows, double* y, int64_t ng) {
    const int nb = (int)((nrows + (1 << LOGBW) - 1) >> LOGBW);
    const int64_t bin_cap = (int64_t)(ng / nb * 1.3) + 4096;
    unsigned* cursor; Ent* bins;
    hipMalloc(&cursor, nb * 4); hipMalloc(&bins, (size_t)nb * bin_cap * sizeof(Ent));
    constexpr int T = SPT * THREADS;
    const int chunks = (int)(S / T);
    const size_t lds = (size_t)T * 18 + nb * 12;
    hipFuncSetAttribute((const void*)k_pb1s<SPT, LOGBW, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float t1 = timeit([&] { hipMemsetAsync(cursor, 0, nb * 4, 0);
        hipLaunchKernelGGL((k_pb1s<SPT, LOGBW, THREADS>), dim3(chunks), dim3(THREADS), lds, 0, keys, vals, nb, bin_cap, cursor, bins); });
    float t12 = timeit([&] { hipMemsetAsync(cursor, 0, nb * 4, 0);
        hipLaunchKernelGGL((k_pb1s<SPT, LOGBW, THREADS>), dim3(chunks), dim3(THREADS), lds, 0, keys, vals, nb, bin_cap, cursor, bins);
        hipLaunchKernelGGL((k_pb2<LOGBW>), dim3(nb), dim3(1024), (8 << LOGBW), 0, cursor, bins, bin_cap, y); });
    printf("(E) PB LDS-sorted: chunk %d slots (%d thr), bin width %d rows (%d bins), LDS %zu B: phase1 %.1f us, both %.1f us  [%s]\n", T, THREADS, 1 << LOGBW, nb, lds, t1, t12,
           hipGetErrorString(hipGetLastError()));
    hipFree(cursor); hipFree(bins);
}

int main() {
    const int64_t S = 1 << 24;                   // slots (C3 capacity)
    const int64_t NX = 1 << 22;                  // up to 32 MB table
    int64_t* keys; double *vals, *x, *out;
    CK(hipMalloc(&keys, S * 8)); CK(hipMalloc(&vals, S * 8)); CK(hipMalloc(&x, NX * 8)); CK(hipMalloc(&out, S * 8));
    std::vector<double> hx(NX, 1.5);
    CK(hipMemcpy(x, hx.data(), NX * 8, hipMemcpyHostToDevice));
    std::vector<double> hv(S, 1.25);
    CK(hipMemcpy(vals, hv.data(), S * 8, hipMemcpyHostToDevice));
    std::vector<int64_t> hk(S);

    // (A)
    const int blocks = 256 * 8 * 4, rounds = 4;
    printf("(A) pure random 8-byte gathers, %d blocks x 256 threads\n", blocks);
    for (int lg = 17; lg <= 22; ++lg) {          // 1 MB .. 32 MB
        const uint32_t mask = (1u << lg) - 1;
        const double ng8 = (double)blocks * 256 * rounds * 8, ng16 = (double)blocks * 256 * rounds * 16, ng4 = (double)blocks * 256 * rounds * 4;
        float t0 = timeit([&] { hipLaunchKernelGGL((k_gather<0, 8>), dim3(blocks), dim3(256), 0, 0, x, mask, rounds, out); });
        float t1 = timeit([&] { hipLaunchKernelGGL((k_gather<1, 8>), dim3(blocks), dim3(256), 0, 0, x, mask, rounds, out); });
        float t2 = timeit([&] { hipLaunchKernelGGL((k_gather<2, 8>), dim3(blocks), dim3(256), 0, 0, x, mask, rounds, out); });
        float t3 = timeit([&] { hipLaunchKernelGGL((k_gather<0, 16>), dim3(blocks), dim3(256), 0, 0, x, mask, rounds, out); });
        float t4 = timeit([&] { hipLaunchKernelGGL((k_gather<0, 4>), dim3(blocks), dim3(256), 0, 0, x, mask, rounds, out); });
        float t5 = timeit([&] { hipLaunchKernelGGL((k_gather<1, 16>), dim3(blocks), dim3(256), 0, 0, x, mask, rounds, out); });
        printf("  table %5.1f MB: plain U8 %6.1f G/s | nt U8 %6.1f | sc1 U8 %6.1f | plain U16 %6.1f | plain U4 %6.1f | nt U16 %6.1f\n",
               (double)(8ull << lg) / 1e6, ng8 / t0 / 1e3, ng8 / t1 / 1e3, ng8 / t2 / 1e3, ng16 / t3 / 1e3, ng4 / t4 / 1e3, ng16 / t5 / 1e3);
    }

    // (B)
    printf("(B) stream of %ld slots x 16 B = %.0f MB\n", (long)S, S * 16 / 1e6);
    for (int64_t i = 0; i < S; ++i) hk[i] = i & 1023;
    CK(hipMemcpy(keys, hk.data(), S * 8, hipMemcpyHostToDevice));
    const int tiles = (int)(S / 2048);
    {
        float a = timeit([&] { hipLaunchKernelGGL((k_stream<1, false>), dim3(tiles), dim3(256), 0, 0, keys, vals, S, out); });
        float b = timeit([&] { hipLaunchKernelGGL((k_stream<1, true>), dim3(tiles), dim3(256), 0, 0, keys, vals, S, out); });
        float c = timeit([&] { hipLaunchKernelGGL((k_stream<2, false>), dim3(tiles), dim3(256), 0, 0, keys, vals, S, out); });
        float d = timeit([&] { hipLaunchKernelGGL((k_stream<2, true>), dim3(tiles), dim3(256), 0, 0, keys, vals, S, out); });
        printf("  8B/lane %.1f us (%.2f TB/s) | 8B nt %.1f us (%.2f) | 16B/lane %.1f us (%.2f) | 16B nt %.1f us (%.2f)\n", a, S * 16 / a / 1e6,
               b, S * 16 / b / 1e6, c, S * 16 / c / 1e6, d, S * 16 / d / 1e6);
    }

    // (C) keys random in [0, nx) with probability 0.6, else -1 (gap)
    for (int lgx = 17; lgx <= 20; ++lgx) {
        const int64_t nx = (lgx == 20) ? 1000000 : (1ll << lgx);
        uint64_t st = 12345;
        int64_t ng = 0;
        for (int64_t i = 0; i < S; ++i) {
            st += 0x9E3779B97F4A7C15ull;
            uint64_t z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            if ((z >> 40) % 1000 < 600) { hk[i] = (int64_t)(z % (uint64_t)nx); ++ng; } else hk[i] = -1;
        }
        CK(hipMemcpy(keys, hk.data(), S * 8, hipMemcpyHostToDevice));
        float a = timeit([&] { hipLaunchKernelGGL((k_both<0, 1>), dim3(tiles), dim3(256), 0, 0, keys, vals, x, out); });
        float b = timeit([&] { hipLaunchKernelGGL((k_both<1, 1>), dim3(tiles), dim3(256), 0, 0, keys, vals, x, out); });
        float c = timeit([&] { hipLaunchKernelGGL((k_both<2, 1>), dim3(tiles), dim3(256), 0, 0, keys, vals, x, out); });
        float d = timeit([&] { hipLaunchKernelGGL((k_both<0, 2>), dim3(tiles), dim3(256), 0, 0, keys, vals, x, out); });
        float e = timeit([&] { hipLaunchKernelGGL((k_both<1, 2>), dim3(tiles), dim3(256), 0, 0, keys, vals, x, out); });
        printf("(C) stream + %.1fM gathers from %.1f MB x: plain/8B %.1f us | nt-x/8B %.1f | sc1-x/8B %.1f | plain/16B %.1f | nt-x/16B %.1f\n",
               ng / 1e6, nx * 8 / 1e6, a, b, c, d, e);

        if (lgx == 20) {
            run_pb<8, 10>(keys, vals, S, nx, out, ng);
            run_pb<8, 12>(keys, vals, S, nx, out, ng);
            run_pb<16, 12>(keys, vals, S, nx, out, ng);
            run_pb<32, 12>(keys, vals, S, nx, out, ng);
            run_pb<16, 13>(keys, vals, S, nx, out, ng);
            run_pb<32, 14>(keys, vals, S, nx, out, ng);

            run_pbs<8, 12, 512>(keys, vals, S, nx, out, ng);
            run_pbs<8, 12, 1024>(keys, vals, S, nx, out, ng);
            run_pbs<4, 12, 1024>(keys, vals, S, nx, out, ng);
            run_pbs<8, 13, 1024>(keys, vals, S, nx, out, ng);
            run_pbs<8, 10, 1024>(keys, vals, S, nx, out, ng);
            run_pbs<8, 12, 256>(keys, vals, S, nx, out, ng);
        }
    }
    printf("done\n");
    return 0;
}
