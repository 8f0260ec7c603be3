// tools/gatherbench4.hip — dev micro-benchmark (not product code): does a tile loop that requests the NEXT span's slot streams before the
// compute phase of the current one recover what the compute phase costs?  The question behind DESIGN.md §8 item 2: k_spmv_gather is 1.3 x
// above the floor of its access pattern where x is L2-resident; a wave spends its life as (stream wait) -> (gather wait) -> (LDS phase +
// walk + stores), and the memory pipeline idles for that wave during the third part.
//   gatherbench4 <log2 slots> <gathers> <x entries>
// Kernels (all key-driven, one wave per 512-slot span, gap lanes read x[0] like the product):
//   P0  the floor: loads -> gathers -> one sum per wave                              (= gatherbench3's key-driven kernel)
//   P1  + a compute phase like the product's: products to LDS, a barrier of the 4 waves, a serial walk of ~32 lanes over ~17 entries
//       each (dependent LDS reads + fp64 adds), one store per lane
//   P2  P1 over T consecutive spans per wave in a loop, no prefetch
//   P3  P2 with the next span's 16 stream loads requested right behind the gathers of the current span
// Build: hipcc --offload-arch=gfx950 -O3 -o gatherbench4 gatherbench4.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int W = 8;            // words (of 64 slots) per span

struct Span { int32_t k[W]; double v[W]; };

__device__ __forceinline__ void load_span(Span& s, const int32_t* __restrict__ kp, const double* __restrict__ vp, int64_t span, int lane) {
    const int64_t b = span * (W * 64) + lane;
#pragma unroll
    for (int j = 0; j < W; ++j) { s.k[j] = __builtin_nontemporal_load(kp + b + j * 64); s.v[j] = __builtin_nontemporal_load(vp + b + j * 64); }
}

// the compute phase of one span: products -> LDS, barrier, every second lane walks 17 entries, one store
__device__ __forceinline__ double compute_phase(const Span& s, const double* xv, double* sP, int lane, bool barrier) {
#pragma unroll
    for (int j = 0; j < W; ++j) sP[j * 64 + lane] = s.k[j] >= 0 ? s.v[j] * xv[j] : 0.0;
    if (barrier) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    double sum = 0.0;
    if ((lane & 1) == 0) {
        const int a = lane * 8;                   // 32 rows of 16 entries
        int t = a;
        for (; t + 3 < a + 17 && t + 3 < W * 64; t += 4) {
            const double t0 = sP[t], t1 = sP[t + 1], t2 = sP[t + 2], t3 = sP[t + 3];
            sum = sum + t0; sum = sum + t1; sum = sum + t2; sum = sum + t3;
        }
        for (; t < a + 17 && t < W * 64; ++t) sum = sum + sP[t];
    }
    return sum;
}

// MODE 0: P0, 1: P1 ; one span per wave
template <int MODE>
__global__ __launch_bounds__(256) void k_one(const int32_t* __restrict__ kp, const double* __restrict__ vp, const double* __restrict__ x,
                                             int nspans, double* __restrict__ y) {
    __shared__ double sPw[4][W * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = nspans / 8;
    const int li = (blockIdx.x >> 3) * 4 + wv;
    if (li >= per) return;
    const int span = (blockIdx.x & 7) * per + li;
    Span s;
    load_span(s, kp, vp, span, lane);
    double xv[W];
#pragma unroll
    for (int j = 0; j < W; ++j) xv[j] = x[s.k[j] >= 0 ? s.k[j] : 0];
    if (MODE == 0) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < W; ++j) acc += s.k[j] >= 0 ? s.v[j] * xv[j] : 0.0;
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) y[span] = acc;
    } else {
        const double sum = compute_phase(s, xv, sPw[wv], lane, true);
        if ((lane & 1) == 0) y[(int64_t)span * 32 + (lane >> 1)] = sum;
    }
}

// T spans per wave (consecutive workgroup-tiles of 4 spans, strided by the number of workgroups of the XCD); PREFETCH: the next span's
// streams are requested right behind the gathers of the current one
template <bool PREFETCH>
__global__ __launch_bounds__(256) void k_loop(const int32_t* __restrict__ kp, const double* __restrict__ vp, const double* __restrict__ x,
                                              int nspans, int T, double* __restrict__ y) {
    __shared__ double sPw[4][W * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = nspans / 8;                    // spans of this XCD
    const int wpx = gridDim.x >> 3;                // workgroups per XCD
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3;
    // iteration i: tile (wg + i * wpx) of the XCD, span = tile * 4 + wv
    auto span_of = [&](int i) { const int tile = wg + i * wpx; const int li = tile * 4 + wv; return li < per ? xcd * per + li : -1; };
    Span cur, nxt;
    int sp = span_of(0);
    if (sp >= 0) load_span(cur, kp, vp, sp, lane);
    for (int i = 0; i < T; ++i) {
        const int spn = i + 1 < T ? span_of(i + 1) : -1;
        double xv[W];
        if (sp >= 0) {
#pragma unroll
            for (int j = 0; j < W; ++j) xv[j] = x[cur.k[j] >= 0 ? cur.k[j] : 0];
        }
        if (PREFETCH && spn >= 0) load_span(nxt, kp, vp, spn, lane);
        double sum = 0.0;
        if (sp >= 0) sum = compute_phase(cur, xv, sPw[wv], lane, false);
        __syncthreads();
        if (sp >= 0 && (lane & 1) == 0) y[(int64_t)sp * 32 + (lane >> 1)] = sum;
        if (!PREFETCH && spn >= 0) load_span(nxt, kp, vp, spn, lane);
        __syncthreads();                            // the slices are rewritten by the next iteration
        cur = nxt; sp = spn;
    }
}


// ---- round 6: what ends a tile in the product — a dependent row-key load per summing lane (the key of the partition whose id sits in LDS) and
// a SCATTERED 8-byte store through it — instead of one coalesced store.  Q1 one span per wave; Q2 the loop; Q3 the loop with the stores of
// tile i issued behind the stream loads and gathers of tile i + 1 (nothing of a tile is waited for before the next tile's requests are out).
__device__ __forceinline__ int rowkey_of(const int* __restrict__ rk, int sp, int lane) { return rk[(int64_t)sp * 32 + (lane >> 1)]; }

__global__ __launch_bounds__(256) void k_one_rk(const int32_t* __restrict__ kp, const double* __restrict__ vp, const double* __restrict__ x,
                                                int nspans, const int* __restrict__ rk, double* __restrict__ y) {
    __shared__ double sPw[4][W * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = nspans / 8;
    const int li = (blockIdx.x >> 3) * 4 + wv;
    if (li >= per) return;
    const int span = (blockIdx.x & 7) * per + li;
    Span s;
    load_span(s, kp, vp, span, lane);
    double xv[W];
#pragma unroll
    for (int j = 0; j < W; ++j) xv[j] = x[s.k[j] >= 0 ? s.k[j] : 0];
    const int row = (lane & 1) == 0 ? rowkey_of(rk, span, lane) : 0;          // requested first, needed last (like the product)
    const double sum = compute_phase(s, xv, sPw[wv], lane, true);
    if ((lane & 1) == 0) y[row] = sum;
}

template <bool DEFER>
__global__ __launch_bounds__(256) void k_loop_rk(const int32_t* __restrict__ kp, const double* __restrict__ vp, const double* __restrict__ x,
                                                 int nspans, int T, const int* __restrict__ rk, double* __restrict__ y) {
    __shared__ double sPw[4][W * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = nspans / 8;
    const int wpx = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3;
    auto span_of = [&](int i) { const int tile = wg + i * wpx; const int li = tile * 4 + wv; return li < per ? xcd * per + li : -1; };
    int prow = -1; double psum = 0.0;                 // DEFER: the store of the previous tile
    for (int i = 0; i < T; ++i) {
        const int sp = span_of(i);
        if (sp < 0) break;                             // (uniform per wave; no workgroup barrier in this kernel)
        Span cur;
        load_span(cur, kp, vp, sp, lane);
        double xv[W];
#pragma unroll
        for (int j = 0; j < W; ++j) xv[j] = x[cur.k[j] >= 0 ? cur.k[j] : 0];
        if (DEFER && prow >= 0) y[prow] = psum;        // behind this tile's requests: nothing waited for it
        const int row = (lane & 1) == 0 ? rowkey_of(rk, sp, lane) : -1;
        const double sum = compute_phase(cur, xv, sPw[wv], lane, false);
        if (DEFER) { prow = row; psum = sum; }
        else if ((lane & 1) == 0) y[row] = sum;
    }
    if (DEFER && prow >= 0) y[prow] = psum;
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
    if (argc < 4) { printf("usage: gatherbench4 <log2 slots> <gathers> <x entries>\n"); return 2; }
    const int lg = atoi(argv[1]);
    const int64_t G = atoll(argv[2]);
    const int64_t NX = atoll(argv[3]);
    if (lg < 16 || lg > 28 || G < 512 || NX < 1 || NX > ((int64_t)1 << 28)) { printf("sizes out of range\n"); return 2; }
    const int64_t S = (int64_t)1 << lg;
    const int nspans = (int)(S / (W * 64));
    int32_t* keys; double *vals, *x, *y;
    CK(hipMalloc(&keys, S * 4)); CK(hipMalloc(&vals, S * 8)); CK(hipMalloc(&x, (size_t)NX * 8)); CK(hipMalloc(&y, (size_t)nspans * 32 * 8));
    CK(hipMemset(vals, 0, S * 8)); CK(hipMemset(x, 0, (size_t)NX * 8));
    {
        std::vector<int32_t> hk((size_t)S);
        uint64_t st = 777;
        const uint64_t thr = (uint64_t)((double)G / (double)S * 4294967296.0);
        for (int64_t i = 0; i < S; ++i) {
            uint64_t z = (st += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            hk[(size_t)i] = (z & 0xffffffffull) < thr ? (int32_t)(((z >> 32) * (uint64_t)NX) >> 32) : -1;
        }
        CK(hipMemcpy(keys, hk.data(), (size_t)S * 4, hipMemcpyHostToDevice));
    }
    printf("slots 2^%d, %d spans, x %.2f MB\n", lg, nspans, NX * 8 / 1e6);
    const int grid1 = (nspans / 8 + 3) / 4 * 8;
    printf("P0 floor (no compute phase)              %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_one<0>, dim3(grid1), dim3(256), 0, 0, keys, vals, x, nspans, y); }));
    printf("P1 + compute phase, one span per wave    %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_one<1>, dim3(grid1), dim3(256), 0, 0, keys, vals, x, nspans, y); }));
    for (int T : {2, 4, 8}) {
        const int tiles_per_xcd = nspans / 8 / 4;
        const int wpx = (tiles_per_xcd + T - 1) / T;
        const int grid = wpx * 8;
        const float a = timeit([&] { hipLaunchKernelGGL(k_loop<false>, dim3(grid), dim3(256), 0, 0, keys, vals, x, nspans, T, y); });
        const float b = timeit([&] { hipLaunchKernelGGL(k_loop<true>, dim3(grid), dim3(256), 0, 0, keys, vals, x, nspans, T, y); });
        printf("T = %d spans per wave (grid %d): P2 loop, no prefetch %.1f us | P3 loop + prefetch %.1f us\n", T, grid, a, b);
    }
    {
        // row keys: span-local ids mapped through a table to rows that are ascending on the whole but scattered within +-2048 (8-byte stores, not coalesced)
        const int64_t NR = (int64_t)nspans * 32;
        std::vector<int> hr((size_t)NR);
        uint64_t st = 4242;
        for (int64_t i = 0; i < NR; i += 4096) {
            const int64_t n = NR - i < 4096 ? NR - i : 4096;
            for (int64_t q = 0; q < n; ++q) hr[(size_t)(i + q)] = (int)(i + q);
            for (int64_t q = n - 1; q > 0; --q) {
                uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 31;
                const int64_t r = (int64_t)(z % (uint64_t)(q + 1));
                std::swap(hr[(size_t)(i + q)], hr[(size_t)(i + r)]);
            }
        }
        int* rk; CK(hipMalloc(&rk, (size_t)NR * 4)); CK(hipMemcpy(rk, hr.data(), (size_t)NR * 4, hipMemcpyHostToDevice));
        printf("Q1 row-key load + scattered store, one span per wave   %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_one_rk, dim3(grid1), dim3(256), 0, 0, keys, vals, x, nspans, rk, y); }));
        for (int T : {2, 4, 8}) {
            const int tiles_per_xcd = nspans / 8 / 4;
            const int wpx = (tiles_per_xcd + T - 1) / T;
            const int grid = wpx * 8;
            const float a = timeit([&] { hipLaunchKernelGGL(k_loop_rk<false>, dim3(grid), dim3(256), 0, 0, keys, vals, x, nspans, T, rk, y); });
            const float b = timeit([&] { hipLaunchKernelGGL(k_loop_rk<true>, dim3(grid), dim3(256), 0, 0, keys, vals, x, nspans, T, rk, y); });
            printf("T = %d (grid %d): Q2 loop %.1f us | Q3 loop, stores deferred behind the next tile's requests %.1f us\n", T, grid, a, b);
        }
        CK(hipFree(rk));
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
