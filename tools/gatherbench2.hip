// tools/gatherbench2.hip — dev micro-benchmark (not product code), round 2: can the slot stream and the x gathers of the
// PCSR SpMV OVERLAP on gfx950, and what do the x-blocked forms cost?
//   slots: 2^24 x (int32 key, f64 value), 65.6 % occupied (key -1 = gap), keys uniform in [0, NX)
//   E0  stream only                       (one wave per 512 slots, or persistent)
//   E1  stream + gather, one span per wave (the round-1 shape)
//   E2  persistent, software-pipelined: gathers of span i are issued, then the stream loads of span i+1, then the
//       gathers are consumed (vmcnt counts in order: the stream stays in flight behind the gathers)
//   E3  two launches, each gathers only keys of one half of x (x half = 4 MB = one XCD L2)
//   E4  one launch, XCDs 0-3 gather from the lower half of x, XCDs 4-7 from the upper half; both groups stream everything
// Build: hipcc --offload-arch=gfx950 -O3 -o gatherbench2 gatherbench2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int W = 8;            // words (of 64 slots) per span

template <bool NT> __device__ __forceinline__ int32_t ldk(const int32_t* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ double ldv(const double* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// MODE 0: stream only; 1: stream + gather.  PIPE: software pipelining depth (0 = none, 1 = next span's stream behind the gathers).
// GROUPS 1: XCD g owns the g-th eighth of the spans; 2: XCD group (g >> 2) gathers only its half of x, each group covers all spans.
template <int MODE, int PIPE, int GROUPS, bool NT>
__global__ __launch_bounds__(256) void k_fused(const int32_t* __restrict__ kp, const double* __restrict__ vp,
                                               const double* __restrict__ x, int64_t nspans, int32_t lo, int32_t hi, int32_t half,
                                               double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7;
    const int64_t wg_in_xcd = blockIdx.x >> 3;
    const int64_t waves_xcd = (int64_t)(gridDim.x >> 3) * 4;
    const int64_t wi = wg_in_xcd * 4 + wv;
    int64_t s, s_end;
    if (GROUPS == 1) {
        const int64_t per = nspans / 8;
        s = xcd * per + wi; s_end = (xcd + 1) * per;
    } else {
        const int64_t per = nspans / 4;
        s = (xcd & 3) * per + wi; s_end = ((xcd & 3) + 1) * per;
        if (xcd >> 2) { lo = half; hi = 2 * half; } else { lo = 0; hi = half; }
    }
    double acc = 0.0;
    if (PIPE == 0) {
        for (; s < s_end; s += waves_xcd) {
            int32_t k[W]; double v[W];
            const int64_t b = s * (W * 64) + lane;
#pragma unroll
            for (int j = 0; j < W; ++j) { k[j] = ldk<NT>(kp + b + j * 64); v[j] = ldv<NT>(vp + b + j * 64); }
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < W; ++j) acc += v[j] + (double)k[j];
            } else {
                double xv[W];
#pragma unroll
                for (int j = 0; j < W; ++j) xv[j] = x[(k[j] >= lo && k[j] < hi) ? k[j] : 0];
#pragma unroll
                for (int j = 0; j < W; ++j) acc += (k[j] >= lo && k[j] < hi) ? v[j] * xv[j] : 0.0;
            }
        }
    } else {
        int32_t k[W]; double v[W];
        if (s < s_end) {
            const int64_t b = s * (W * 64) + lane;
#pragma unroll
            for (int j = 0; j < W; ++j) { k[j] = ldk<NT>(kp + b + j * 64); v[j] = ldv<NT>(vp + b + j * 64); }
        }
        for (; s < s_end; s += waves_xcd) {
            double xv[W];
#pragma unroll
            for (int j = 0; j < W; ++j) xv[j] = x[(k[j] >= lo && k[j] < hi) ? k[j] : 0];       // clamped, not predicated
            __builtin_amdgcn_sched_barrier(0);
            int32_t k2[W]; double v2[W];
            const int64_t sn = s + waves_xcd < s_end ? s + waves_xcd : s;       // clamped: straight-line issue
            const int64_t b = sn * (W * 64) + lane;
#pragma unroll
            for (int j = 0; j < W; ++j) { k2[j] = ldk<NT>(kp + b + j * 64); v2[j] = ldv<NT>(vp + b + j * 64); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < W; ++j) acc += (k[j] >= lo && k[j] < hi) ? v[j] * xv[j] : 0.0;
#pragma unroll
            for (int j = 0; j < W; ++j) { k[j] = k2[j]; v[j] = v2[j]; }
        }
    }
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}


// E5: do stream work and gather work overlap when they run on DIFFERENT CUs / XCDs / waves?  Two work queues (atomic
// cursors): queue 0 = stream spans (12 B slots), queue 1 = gather spans (512 hashed gathers each, no other traffic).
// A wave starts on the queue of its role and helps with the other one when its own is empty.
// ROLE 0: cu_id & 1   1: XCC id >> 2   2: wave parity in the workgroup   3: everybody starts on the stream queue
__device__ __forceinline__ uint64_t mixh(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <int ROLE>
__global__ __launch_bounds__(256) void k_roles(const int32_t* __restrict__ kp, const double* __restrict__ vp, const double* __restrict__ x,
                                               uint32_t mask, int nstream, int ngather, int* __restrict__ cursors, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    double acc = 0.0;
    // static assignments: (first, step) per queue; ROLE 0 uses chunked atomic cursors per (queue, XCD)
    for (int q = 0; q < 2; ++q) {
        int first = 0, step = 1, lim = q ? ngather : nstream;
        bool dynamic = false;
        if (ROLE == 0) { if ((int)((hw >> 8) & 1) != q) continue; dynamic = true; }
        else if (ROLE == 1) { if ((int)(xcc >> 2) != q) continue; const int per = lim / 4; first = (xcc & 3) * per + (blockIdx.x >> 3) * 4 + wv; step = (gridDim.x >> 3) * 4; lim = ((xcc & 3) + 1) * per; }
        else if (ROLE == 2) { if ((wv & 1) != q) continue; first = blockIdx.x * 2 + (wv >> 1); step = gridDim.x * 2; }
        else { first = blockIdx.x * 4 + wv; step = gridDim.x * 4; }
        int s = first, chunk_left = 0;
        const int per8 = lim / 8;
        while (true) {
            if (dynamic) {
                if (chunk_left == 0) {
                    int c = 0;
                    if (lane == 0) c = atomicAdd(&cursors[(q * 8 + xcc) * 32], 8);
                    s = __builtin_amdgcn_readfirstlane(c);
                    if (s >= per8) break;
                    chunk_left = per8 - s < 8 ? per8 - s : 8;
                    s += xcc * per8;
                }
                --chunk_left;
            } else if (s >= lim) break;
            if (q == 0) {
                int32_t k[W]; double v[W];
                const int64_t b = (int64_t)s * (W * 64) + lane;
#pragma unroll
                for (int j = 0; j < W; ++j) { k[j] = kp[b + j * 64]; v[j] = vp[b + j * 64]; }
#pragma unroll
                for (int j = 0; j < W; ++j) acc += v[j] + (double)k[j];
            } else {
                double t[W];
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const uint32_t idx = (uint32_t)mixh(((uint64_t)s * 64 + lane) * 1315423911ull + (uint64_t)j * 0x9E3779B97F4A7C15ull) & mask;
                    t[j] = x[idx];
                }
#pragma unroll
                for (int j = 0; j < W; ++j) acc += t[j];
            }
            s += dynamic ? 1 : step;
        }
    }
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
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

static uint64_t sm_state = 12345;
static uint64_t splitmix() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const bool prof = argc > 1 && std::string(argv[1]) == "prof";
    const int64_t S = 1 << 24;
    const int64_t nspans = S / (W * 64);
    int32_t* keys; double *vals, *x, *out;
    CK(hipMalloc(&keys, S * 4)); CK(hipMalloc(&vals, S * 8)); CK(hipMalloc(&x, (size_t)(1 << 21) * 8)); CK(hipMalloc(&out, S * 8));
    std::vector<double> hv(S, 1.5), hx(1 << 21, 1.25);
    CK(hipMemcpy(vals, hv.data(), S * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    std::vector<int32_t> hk(S);
    const int64_t nonpers = nspans / 4;       // workgroups when every wave takes exactly one span
    printf("slots %lld, spans %lld, stream bytes %.1f MB\n", (long long)S, (long long)nspans, S * 12 / 1e6);

    if (prof) {
        // rocprofv3 --pmc mode: a fixed list of dispatches (3 of each), summarised per dispatch index by tools/scripts/pmc_gatherbench2.sh
        auto fill = [&](int64_t NX) {
            sm_state = 777;
            for (int64_t i = 0; i < S; ++i) {
                const uint64_t r = splitmix();
                hk[i] = ((r & 0xffff) < 42991) ? (int32_t)((r >> 20) % (uint64_t)NX) : -1;
            }
            return hipMemcpy(keys, hk.data(), S * 4, hipMemcpyHostToDevice);
        };
        CK(fill(1000000));
        const int32_t nx = 1000000, half = 500000;
        int d = 0;
        auto L = [&](const char* name) { printf("dispatch %d: %s\n", d++, name); };
        for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL((k_fused<0, 0, 1, true>), dim3((unsigned)nonpers), dim3(256), 0, 0, keys, vals, x, nspans, 0, nx, half, out); L("E0 stream only (12 B slots, nt)"); }
        for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL((k_fused<1, 0, 1, true>), dim3((unsigned)nonpers), dim3(256), 0, 0, keys, vals, x, nspans, 0, nx, half, out); L("E1 single pass, x 8 MB"); }
        for (int r = 0; r < 3; ++r) {
            hipLaunchKernelGGL((k_fused<1, 0, 1, true>), dim3((unsigned)nonpers), dim3(256), 0, 0, keys, vals, x, nspans, 0, half, half, out); L("E3 pass over the lower x half");
            hipLaunchKernelGGL((k_fused<1, 0, 1, true>), dim3((unsigned)nonpers), dim3(256), 0, 0, keys, vals, x, nspans, half, nx, half, out); L("E3 pass over the upper x half");
        }
        for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL((k_fused<1, 0, 2, true>), dim3(2048), dim3(256), 0, 0, keys, vals, x, nspans, 0, nx, half, out); L("E4 two XCD groups, one launch (persistent, 2048 wg)"); }
        CK(fill(131072));
        for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL((k_fused<1, 0, 1, false>), dim3((unsigned)nonpers), dim3(256), 0, 0, keys, vals, x, nspans, 0, 131072, 65536, out); L("E1 single pass, x 1 MB (L2-resident), plain stream"); }
        CK(hipDeviceSynchronize());
        return 0;
    }
    const int64_t NXs[] = {131072, 262144, 500000, 1000000};
    for (int64_t NX : NXs) {
        sm_state = 777;
        int64_t cells = 0;
        for (int64_t i = 0; i < S; ++i) {
            const uint64_t r = splitmix();
            if ((r & 0xffff) < 42991) { hk[i] = (int32_t)((r >> 20) % (uint64_t)NX); ++cells; } else hk[i] = -1;
        }
        CK(hipMemcpy(keys, hk.data(), S * 4, hipMemcpyHostToDevice));
        printf("== x %.2f MB (%lld entries), %.2f M cells\n", NX * 8 / 1e6, (long long)NX, cells / 1e6);
        const int32_t nx = (int32_t)NX, half = (int32_t)(NX / 2);
#define RUN(MODE, PIPE, GROUPS, NT, GRID, LO, HI) \
        timeit([&] { hipLaunchKernelGGL((k_fused<MODE, PIPE, GROUPS, NT>), dim3((unsigned)(GRID)), dim3(256), 0, 0, keys, vals, x, nspans, LO, HI, half, out); })
        if (NX == NXs[0]) {
            printf("E0 stream only: one span/wave nt %.1f us | plain %.1f | persistent 2048 wg nt %.1f | 1024 wg %.1f | 4096 wg %.1f\n",
                   RUN(0, 0, 1, true, nonpers, 0, nx), RUN(0, 0, 1, false, nonpers, 0, nx), RUN(0, 0, 1, true, 2048, 0, nx),
                   RUN(0, 0, 1, true, 1024, 0, nx), RUN(0, 0, 1, true, 4096, 0, nx));
        }
        printf("E1 fused, one span per wave: nt %.1f us | plain stream %.1f\n", RUN(1, 0, 1, true, nonpers, 0, nx), RUN(1, 0, 1, false, nonpers, 0, nx));
        printf("E1p fused, persistent unpipelined: 1024 wg %.1f | 2048 wg %.1f | 4096 wg %.1f\n",
               RUN(1, 0, 1, true, 1024, 0, nx), RUN(1, 0, 1, true, 2048, 0, nx), RUN(1, 0, 1, true, 4096, 0, nx));
        printf("E2 fused, persistent pipelined:   1024 wg %.1f | 2048 wg %.1f | 4096 wg %.1f | 8192 wg %.1f\n",
               RUN(1, 1, 1, true, 1024, 0, nx), RUN(1, 1, 1, true, 2048, 0, nx), RUN(1, 1, 1, true, 4096, 0, nx), RUN(1, 1, 1, true, 8192, 0, nx));
        if (NX == 1000000) {
            auto two = [&](auto f0, auto f1) { return timeit([&] { f0(); f1(); }); };
            float a = two([&] { hipLaunchKernelGGL((k_fused<1, 0, 1, true>), dim3((unsigned)nonpers), dim3(256), 0, 0, keys, vals, x, nspans, 0, half, half, out); },
                          [&] { hipLaunchKernelGGL((k_fused<1, 0, 1, true>), dim3((unsigned)nonpers), dim3(256), 0, 0, keys, vals, x, nspans, half, nx, half, out); });
            float b = two([&] { hipLaunchKernelGGL((k_fused<1, 1, 1, true>), dim3(2048), dim3(256), 0, 0, keys, vals, x, nspans, 0, half, half, out); },
                          [&] { hipLaunchKernelGGL((k_fused<1, 1, 1, true>), dim3(2048), dim3(256), 0, 0, keys, vals, x, nspans, half, nx, half, out); });
            float c = two([&] { hipLaunchKernelGGL((k_fused<1, 1, 1, true>), dim3(4096), dim3(256), 0, 0, keys, vals, x, nspans, 0, half, half, out); },
                          [&] { hipLaunchKernelGGL((k_fused<1, 1, 1, true>), dim3(4096), dim3(256), 0, 0, keys, vals, x, nspans, half, nx, half, out); });
            printf("E3 two launches over x halves: one span/wave %.1f us | pipelined 2048 wg %.1f | pipelined 4096 wg %.1f\n", a, b, c);
            float q = 0;
            {
                const int32_t qd = nx / 4;
                q = timeit([&] { for (int p = 0; p < 4; ++p) hipLaunchKernelGGL((k_fused<1, 1, 1, true>), dim3(2048), dim3(256), 0, 0, keys, vals, x, nspans, p * qd, (p + 1) * qd, half, out); });
            }
            printf("E3q four launches over x quarters (pipelined 2048 wg): %.1f us\n", q);
            printf("E4 one launch, two XCD groups: one span/wave-ish (grid %lld) %.1f us | pipelined 2048 wg %.1f | 4096 wg %.1f | unpipelined 2048 wg %.1f\n",
                   (long long)(nonpers * 2), RUN(1, 0, 2, true, nonpers * 2, 0, nx), RUN(1, 1, 2, true, 2048, 0, nx), RUN(1, 1, 2, true, 4096, 0, nx),
                   RUN(1, 0, 2, true, 2048, 0, nx));
        }
    }

    {
        int* cursors; CK(hipMalloc(&cursors, 16 * 32 * 4));
        const int ngather = (int)(nspans * 0.656);
        for (uint32_t mask : {0x1ffffu, 0xfffffu}) {
            printf("== E5 roles, x table %.2f MB: %d stream spans, %d gather spans\n", (mask + 1) * 8 / 1e6, (int)nspans, ngather);
#define ROLES(R, NS, NG, GRID) timeit([&] { hipMemsetAsync(cursors, 0, 16 * 32 * 4, 0); hipLaunchKernelGGL((k_roles<R>), dim3(GRID), dim3(256), 0, 0, keys, vals, x, mask, NS, NG, cursors, out); })
            for (int grid : {1024, 2048, 4096}) {
                const int NS = (int)nspans;
                printf("  grid %d: every wave both queues: stream only %.1f us | gather only %.1f | both %.1f\n", grid, ROLES(3, NS, 0, grid), ROLES(3, 0, ngather, grid), ROLES(3, NS, ngather, grid));
                printf("           by wave parity:       stream only %.1f us | gather only %.1f | both %.1f\n", ROLES(2, NS, 0, grid), ROLES(2, 0, ngather, grid), ROLES(2, NS, ngather, grid));
                printf("           by CU parity:         stream only %.1f us | gather only %.1f | both %.1f\n", ROLES(0, NS, 0, grid), ROLES(0, 0, ngather, grid), ROLES(0, NS, ngather, grid));
                printf("           by XCD half:          stream only %.1f us | gather only %.1f | both %.1f\n", ROLES(1, NS, 0, grid), ROLES(1, 0, ngather, grid), ROLES(1, NS, ngather, grid));
            }
        }
    }
    printf("done\n");
    return 0;
}
