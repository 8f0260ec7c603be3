// csrc/sparsex.hip — sparse-x product  mat * v  with the exact result shape of the reference's _mul (src/operations.jl:107-135,
// _mul_output :11-12): the TOUCHED rows only, ascending, stored zeros kept.  This is the product Coluna calls.
//
// Round 6 form.  Two persistent arrays per matrix with a ZERO INVARIANT (all zero between two products): acc[ny] (Float64 sums) and
// bm[ny / 64] (one bit per touched row).  A product is three launches, no memset, no copy command in between:
//   k_spx_accum   one wave per stored x entry: locate the column (direct hit when the column table is the identity up to the entry,
//                 else the 64-ary search), walk its slot range, acc[row] += x_j * a with fp64 atomics, bm |= bit(row)
//   k_spx_count   touched rows per 4096-row tile from the bitmap; the workgroup that finishes last (ticket) turns the counts into
//                 exclusive prefixes and the total
//   k_spx_emit    (row, acc[row]) pairs in ascending row order into the packed result (HBM; the first cells also straight into a
//                 pinned landing area, with the count and a sequence number the host polls for), and acc / bm zeroed again on the way
// Work is proportional to the matched cells + ny / 64 bitmap words, not to ny doubles (rounds 1-5: two memsets of ny doubles / bytes,
// a byte -> bitmap pass, three compaction launches and a stream synchronisation per product).
// The dense-ish branch (many stored entries: gather kernel over the twin orientation) hands its y / pattern vectors to the same count
// and emit kernels.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <mutex>
#include "dsa_dev.h"

namespace dsa {

namespace {
__device__ __forceinline__ uint32_t sx_wave_reduce_add(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t sx_wave_excl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
    return inc - v;
}
}  // namespace

// ---- accumulate ---------------------------------------------------------------------------------------------------------------
// x entries come from HBM or straight from pinned host memory (few entries: no copy command in front of the launch).
__global__ __launch_bounds__(256) void k_spx_accum(KeyArr keys, const double* __restrict__ vals, const uint64_t* __restrict__ occ, int64_t capacity,
                                                   const int64_t* __restrict__ sems, const int64_t* __restrict__ col_keys,
                                                   const uint8_t* __restrict__ col_live, int64_t table_len,
                                                   const int64_t* __restrict__ xi, const double* __restrict__ xv, int64_t nx,
                                                   double* __restrict__ acc, unsigned long long* __restrict__ bm, int64_t ny) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= nx) return;
    const int64_t col = xi[e];
    const double xval = xv[e];
    // Direct hit first: a table without holes in front of `col` has the column at index col (ids are positions, keys ascend: a
    // matrix whose columns are 1..n, the usual case).  Everything the hit needs is requested in ONE round: key, liveness, the
    // column's semaphore and the next one.
    int64_t pos = 0, from = 0, to = 0;
    bool have = false;
    if (col >= 1 && col <= table_len) {
        const int64_t ck = col_keys[col - 1];
        const uint8_t lv = col_live[col - 1];
        const int64_t s0 = sems[col - 1];
        const int64_t s1 = col < table_len ? sems[col] : 0;
        if (ck == col && lv && s0 != 0) {
            pos = col; from = s0 + 1; have = true;
            if (col == table_len) to = capacity;
            else if (s1 != 0) to = s1 - 1;
            else have = false;                      // a tombstone behind it: the general walk below
        }
    }
    if (!have) {
        // largest live table index with col_keys <= col  (64-ary narrowing; live keys ascend with the index)
        int64_t L = 0, H = table_len;
        while (H - L > 64) {
            const int64_t width = H - L;
            const int64_t p = L + (width * (lane + 1)) / 64;
            int64_t q = p;
            while (q > L && !col_live[q - 1]) --q;
            bool pr = true;
            if (q > L) pr = col_keys[q - 1] <= col;
            const uint64_t nb = ~__ballot(pr);
            const int j = nb ? __ffsll((unsigned long long)nb) - 1 : 64;
            const int64_t pj = L + (width * (j + 1)) / 64;
            const int64_t pj1 = L + (width * j) / 64;
            if (j < 64) H = pj - 1;
            L = pj1;
        }
        const int64_t p = L + 1 + lane;
        bool viol = false;
        if (p <= H && col_live[p - 1]) viol = col_keys[p - 1] > col;
        const uint64_t b = __ballot(viol);
        pos = b ? L + __ffsll((unsigned long long)b) - 1 : H;
        while (pos > 0 && !col_live[pos - 1]) --pos;
        if (pos == 0 || col_keys[pos - 1] != col) return;            // no such column: x entry skipped (src/operations.jl:76-79)
        from = sems[pos - 1] + 1;
        int64_t nxt = pos + 1;
        while (nxt <= table_len && sems[nxt - 1] == 0) ++nxt;
        to = nxt <= table_len ? sems[nxt - 1] - 1 : capacity;
    }
    for (int64_t s = from + lane; s <= to; s += 64) {
        if ((occ[(s - 1) >> 6] >> ((s - 1) & 63)) & 1ull) {
            const int64_t row = keys[s - 1];
            if (row >= 1 && row <= ny) {
                atomicAdd(&acc[row - 1], xval * vals[s - 1]);
                atomicOr(&bm[(row - 1) >> 6], 1ull << ((row - 1) & 63));
            }
        }
    }
}

// ---- touched rows of a gather product: bit r <=> pattern[r] != 0 (the 0/1 pass of the dense-ish branch) ---------------------------
__global__ __launch_bounds__(256) void k_spx_pattern_bits(const double* __restrict__ pattern, int64_t ny, unsigned long long* __restrict__ bm, int64_t nwords) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nwords) return;
    const int64_t r = (w << 6) + lane;
    const bool t = r < ny && pattern[r] != 0.0;
    const uint64_t b = __ballot(t);
    if (lane == 0) bm[w] = b;
}

// ---- count: tile = 64 bitmap words = 4096 rows; scratch = tile_cnt[ntiles], tile_off[ntiles + 1], ticket ---------------------------
constexpr int SX_CNT_TILES = 16;          // tiles per workgroup (4 waves x 4)
__global__ __launch_bounds__(256) void k_spx_count(const unsigned long long* __restrict__ bm, int64_t nwords, int64_t ntiles,
                                                   uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ tile_off, unsigned int* __restrict__ ticket) {
    __shared__ unsigned int sLast;
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t pc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = (int64_t)blockIdx.x * SX_CNT_TILES + wv * 4 + i;
        const int64_t w = t * 64 + lane;
        pc[i] = 0;
        if (t < ntiles && w < nwords) pc[i] = (uint32_t)__popcll(__hip_atomic_load(bm + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = (int64_t)blockIdx.x * SX_CNT_TILES + wv * 4 + i;
        const uint32_t r = sx_wave_reduce_add(pc[i]);
        if (lane == 0 && t < ntiles) __hip_atomic_store(tile_cnt + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sLast = t == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!sLast) return;
    // the last workgroup: exclusive prefix of the tile counts, 1024 per round (4 per thread)
    uint32_t carry = 0;
    for (int64_t base = 0; base < ntiles; base += 1024) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 4;
        uint32_t c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = i0 + i < ntiles ? __hip_atomic_load(tile_cnt + i0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t local = c[0] + c[1] + c[2] + c[3];
        const uint32_t ex = sx_wave_excl_scan(local);
        if (lane == 63) wsum[wv] = ex + local;
        __syncthreads();
        uint32_t run = carry + ex, total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t wsk = wsum[k]; if (k < wv) run += wsk; total += wsk; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (i0 + i < ntiles) tile_off[i0 + i] = run; run += c[i]; }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { tile_off[ntiles] = carry; __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// ---- emit: one wave per tile; values from `src` (acc, cleared on the way when clear_src; or the y of a gather product) --------------
// out_i / out_v: packed result in HBM (cap entries: pairs beyond it are dropped, the count still says how many there are);
// d_count: the count in HBM; host (pinned, may be null): [0] count, [1] sequence number, then pin_cells rows and pin_cells values.
// The workgroup that finishes last publishes the count (everybody's pinned cells have left by then).
__global__ __launch_bounds__(64) void k_spx_emit(unsigned long long* __restrict__ bm, int64_t nwords, int64_t ntiles, const uint32_t* __restrict__ tile_off,
                                                 double* __restrict__ src, int clear_src, int64_t* __restrict__ out_i, double* __restrict__ out_v,
                                                 int64_t cap, int64_t* __restrict__ d_count, long long* __restrict__ host, int64_t pin_cells,
                                                 unsigned long long seq, unsigned int* __restrict__ ticket) {
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    const int64_t wl = t * 64 + lane;
    const uint64_t myword = wl < nwords ? bm[wl] : 0ull;
    const uint32_t myoff = sx_wave_excl_scan((uint32_t)__popcll(myword));
    const int64_t base = tile_off[t];
    uint64_t nz = __ballot(myword != 0ull);
    while (nz) {
        const int w = __ffsll((unsigned long long)nz) - 1;
        nz &= nz - 1;
        const uint64_t mask = __shfl(myword, w, 64);
        const int64_t woff = base + (int64_t)__shfl(myoff, w, 64);
        if ((mask >> lane) & 1ull) {
            const int64_t r = woff + __popcll(mask & mask_lt(lane));
            const int64_t row0 = ((t * 64 + w) << 6) + lane;
            const double v = src[row0];
            if (clear_src) src[row0] = 0.0;
            if (r < cap) { out_i[r] = row0 + 1; out_v[r] = v; }
            if (host != nullptr && r < pin_cells) {
                __hip_atomic_store(host + 8 + r, (long long)(row0 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(host + 8 + pin_cells + r, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (myword != 0ull && wl < nwords) bm[wl] = 0ull;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        const unsigned int k = __hip_atomic_fetch_add(ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == gridDim.x - 1) {
            const long long total = (long long)tile_off[ntiles];
            if (d_count != nullptr) *d_count = total;
            __hip_atomic_store(ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (host != nullptr) {
                __hip_atomic_store(host + 0, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __atomic_thread_fence(__ATOMIC_RELEASE);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(host) + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

hipError_t launch_spx_accum(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, const int64_t* sems, const int64_t* col_keys,
                            const uint8_t* col_live, int64_t table_len, const int64_t* xi, const double* xv, int64_t nx, double* acc,
                            uint64_t* bm, int64_t ny, hipStream_t stream) {
    if (nx <= 0 || ny <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_spx_accum, dim3((unsigned)((nx + 3) / 4)), dim3(256), 0, stream, keys, vals, occ, capacity, sems, col_keys, col_live,
                       table_len, xi, xv, nx, acc, reinterpret_cast<unsigned long long*>(bm), ny);
    return hipGetLastError();
}
hipError_t launch_spx_pattern_bits(const double* pattern, int64_t ny, uint64_t* bm, hipStream_t stream) {
    const int64_t nwords = (ny + 63) >> 6;
    if (nwords <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_spx_pattern_bits, dim3((unsigned)((nwords + 3) / 4)), dim3(256), 0, stream, pattern, ny, reinterpret_cast<unsigned long long*>(bm), nwords);
    return hipGetLastError();
}
// count + emit (a single-workgroup count + emit for small products was measured: 30 vs 15 us per product — dropped); scratch = the caller's (tile_cnt, tile_off [ntiles + 1], ticket [2 words, zero between launches])
hipError_t launch_spx_finish(uint64_t* bm, int64_t ny, uint32_t* tile_cnt, uint32_t* tile_off, unsigned int* ticket, double* src, int clear_src,
                             int64_t* out_i, double* out_v, int64_t cap, int64_t* d_count, long long* host, int64_t pin_cells,
                             unsigned long long seq, hipStream_t stream) {
    const int64_t nwords = (ny + 63) >> 6;
    const int64_t ntiles = (nwords + 63) / 64;
    if (ntiles <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_spx_count, dim3((unsigned)((ntiles + SX_CNT_TILES - 1) / SX_CNT_TILES)), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned long long*>(bm), nwords, ntiles, tile_cnt, tile_off, ticket);
    hipLaunchKernelGGL(k_spx_emit, dim3((unsigned)ntiles), dim3(64), 0, stream, reinterpret_cast<unsigned long long*>(bm), nwords, ntiles,
                       (const uint32_t*)tile_off, src, clear_src, out_i, out_v, cap, d_count, host, pin_cells, seq, ticket);
    return hipGetLastError();
}

}  // namespace dsa
