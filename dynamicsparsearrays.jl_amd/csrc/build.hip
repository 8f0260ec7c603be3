// csrc/build.hip — K-build: the bulk (fill-mode flush) constructor on the device.
//
// Reproduces _dynamicsparse (src/pcsr.jl:354-431) + the PackedCSC constructor's cell stream (src/pcsr.jl:26-63) for one
// orientation: sort the (partition, key) pairs, combine duplicates, count the cells per partition and emit the ordered stream
// [sem(0, id), entries...] per partition, plus the partition keys.  The stream is written straight into the PMA's slot buffer;
// the full-array spread (src/pma.jl:42-55) is then one k_move2<PACKED> launch (rebalance.hip).
//
//   sort   : ONE hand-written stable LSD radix sort (8-bit digits) of 64-bit composites
//                comp = (partition - pmin) << kbits | (key - kmin)
//            carrying the Float64 value — only over the bits the composite really has: kbits + pbits, from a min / max pass
//            over the input (config 3: 20 + 20 bits = 5 passes; round 2 ran 2 x 16 passes of a 64-bit library sort plus two index
//            gathers).  Stable, so equal (partition, key) pairs stay in INPUT order: the reference sorts with an unstable QuickSort
//            (src/pcsr.jl:360); input order is one of its legal outcomes and makes the Float64 fold of duplicates deterministic.
//            A pass is three launches: per-tile digit histogram (LDS atomics), one workgroup per digit scans its row of the
//            histogram matrix (no look-back chain: see DESIGN §3.1 on what chained scans cost on this part), and the scatter:
//            ranks by wave-wide digit matching (8 ballots), the tile sorted through LDS so that runs of equal digits leave as
//            contiguous stores.
//   flags  : new-partition / new-cell flags from neighbour compares of the sorted composites; per-tile counts, one small scan.
//   emit   : every first-of-run lane folds its duplicate run left to right (src/pcsr.jl:374-375) and writes its cell at
//            rank-1 + partition_id ; every first-of-partition lane writes the semaphore cell (0, id) at rank-1 + id-1 and the
//            partition key.  Runs longer than 64 duplicates are finished by a wave each (k_fold_long): coalesced loads, the same
//            left-to-right order.
// Composites that do not fit 64 bits (partition AND key ranges beyond 2^32) take the general path at the end of the file: two
// stable 64-bit sorts by key then by partition (rocPRIM) carrying the input index.
// Bound: HBM (sort passes: 40 B per triple and pass).
#include "dsa_dev.h"
#include <functional>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

namespace dsa {

constexpr int RS_BLOCK = 256, RS_WAVES = RS_BLOCK / 64, RS_ITEMS = 16, RS_TILE = RS_BLOCK * RS_ITEMS, RS_BINS = 256;
constexpr int RS_PER_WAVE = RS_TILE / RS_WAVES;
constexpr int FOLD_INLINE = 64;            // duplicates of one cell folded by the emitting lane itself

// device-resident scalars of one build
struct BuildCtl {
    long long pmin, pmax, kmin, kmax;      // k_minmax
    unsigned long long ncells, nparts;     // k_bf_scan
    unsigned long long nlong;              // runs handed to k_fold_long
    unsigned long long zeros;              // k_minmax: bit 0 = a partition key is 0, bit 1 = a key is 0 (the reserved semaphore key)
};
struct LongRun { int64_t next; int64_t pos; double acc; uint64_t comp; int64_t dpos; };

// min / max of the key arrays when the caller does not know them: per-workgroup partials, folded by the workgroup that finishes
// last (64-bit signed atomic min / max from thousands of waves on four words cost 450 us at 10 M triples)
constexpr int MM_BLOCKS = 1024;
__global__ __launch_bounds__(256) void k_minmax(const int64_t* __restrict__ part, const int64_t* __restrict__ key, int64_t n, BuildCtl* c,
                                                long long* __restrict__ partial /*[4 * gridDim]*/, unsigned int* __restrict__ ticket /*[2]: ticket, zero flags*/) {
    unsigned int* zflags = ticket + 1;
    __shared__ long long sM[4][4];
    __shared__ unsigned int sLast;
    long long m[4] = {INT64_MAX, INT64_MIN, INT64_MAX, INT64_MIN};      // pmin, pmax, kmin, kmax
    unsigned int z = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const long long k = key[i];
        m[2] = k < m[2] ? k : m[2]; m[3] = k > m[3] ? k : m[3];
        if (k == 0) z |= 2u;
        if (part != nullptr) { const long long p = part[i]; m[0] = p < m[0] ? p : m[0]; m[1] = p > m[1] ? p : m[1]; if (p == 0) z |= 1u; }
    }
    if (__ballot(z & 1u)) z |= 1u;
    if (__ballot(z & 2u)) z |= 2u;
    if (z && (threadIdx.x & 63) == 0) atomicOr(zflags, z);
    auto fold = [&](long long* v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const long long a = __shfl_xor(v[q], o, 64); v[q] = (q & 1) ? (a > v[q] ? a : v[q]) : (a < v[q] ? a : v[q]); }
        }
    };
    fold(m);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) for (int q = 0; q < 4; ++q) sM[wv][q] = m[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) for (int q = 0; q < 4; ++q) m[q] = (q & 1) ? (sM[w][q] > m[q] ? sM[w][q] : m[q]) : (sM[w][q] < m[q] ? sM[w][q] : m[q]);
        for (int q = 0; q < 4; ++q) __hip_atomic_store(partial + 4 * blockIdx.x + q, m[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        sLast = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!sLast) return;
    m[0] = m[2] = INT64_MAX; m[1] = m[3] = INT64_MIN;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256)
        for (int q = 0; q < 4; ++q) {
            const long long a = __hip_atomic_load(partial + 4 * b + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m[q] = (q & 1) ? (a > m[q] ? a : m[q]) : (a < m[q] ? a : m[q]);
        }
    fold(m);
    __syncthreads();
    if (lane == 0) for (int q = 0; q < 4; ++q) sM[wv][q] = m[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) for (int q = 0; q < 4; ++q) m[q] = (q & 1) ? (sM[w][q] > m[q] ? sM[w][q] : m[q]) : (sM[w][q] < m[q] ? sM[w][q] : m[q]);
        c->pmin = m[0]; c->pmax = m[1]; c->kmin = m[2]; c->kmax = m[3];
        c->zeros = __hip_atomic_load(zflags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *ticket = 0; *zflags = 0;
    }
}

// the same ranges for a STREAM of triples that arrives in pieces (the device-resident fill buffer): every piece is folded into five
// running words acc = {amin, amax, bmin, bmax, zero flags} behind its upload — a handful of workgroups, one atomic per workgroup and
// word — so that closefillmode! finds the ranges of everything appended without another pass over 160 MB of keys
__global__ __launch_bounds__(256) void k_minmax_acc(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t n, long long* __restrict__ acc) {
    __shared__ long long sM[4][4];
    __shared__ unsigned int sZ;
    long long m[4] = {INT64_MAX, INT64_MIN, INT64_MAX, INT64_MIN};
    unsigned int z = 0;
    if (threadIdx.x == 0) sZ = 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const long long x = a[i], y = b[i];
        m[0] = x < m[0] ? x : m[0]; m[1] = x > m[1] ? x : m[1];
        m[2] = y < m[2] ? y : m[2]; m[3] = y > m[3] ? y : m[3];
        if (x == 0) z |= 1u;
        if (y == 0) z |= 2u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const long long t = __shfl_xor(m[q], o, 64); m[q] = (q & 1) ? (t > m[q] ? t : m[q]) : (t < m[q] ? t : m[q]); }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) for (int q = 0; q < 4; ++q) sM[wv][q] = m[q];
    if (z) atomicOr(&sZ, z);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) for (int q = 0; q < 4; ++q) m[q] = (q & 1) ? (sM[w][q] > m[q] ? sM[w][q] : m[q]) : (sM[w][q] < m[q] ? sM[w][q] : m[q]);
        atomicMin(acc + 0, m[0]); atomicMax(acc + 1, m[1]); atomicMin(acc + 2, m[2]); atomicMax(acc + 3, m[3]);
        if (sZ) atomicOr(reinterpret_cast<unsigned long long*>(acc + 4), (unsigned long long)sZ);
    }
}
hipError_t launch_key_scan_acc(const int64_t* d_a, const int64_t* d_b, int64_t n, long long* d_acc, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 4095) / 4096, 64);
    hipLaunchKernelGGL(k_minmax_acc, dim3(blocks), dim3(256), 0, stream, d_a, d_b, n, d_acc);
    return hipGetLastError();
}

// ---- the radix pass ---------------------------------------------------------------------------------------------------
// element order inside a tile: wave w owns RS_PER_WAVE consecutive elements, iteration j of the wave 64 consecutive ones
__device__ __forceinline__ int64_t rs_index(int64_t tile0, int wv, int j, int lane) { return tile0 + (int64_t)wv * RS_PER_WAVE + j * 64 + lane; }

// composites from (partition, key) + the histogram of the first pass: hist[d * nblocks + b]
// ibits > 0: the element's input index rides in the low ibits bits of the word (composite << ibits | index): the sort then moves
// 8-byte words only — no value payload, values are gathered by index when the cells are emitted — and, being stable over the
// composite's bits alone (shift starts at ibits), keeps equal composites in input order
__global__ __launch_bounds__(RS_BLOCK) void k_comp_hist(const int64_t* __restrict__ part, const int64_t* __restrict__ key, int64_t n,
                                                        int64_t pmin, int64_t kmin, int kbits, int ibits, uint64_t* __restrict__ comp,
                                                        int shift, uint32_t* __restrict__ hist, int64_t nblocks) {
    __shared__ uint32_t h[RS_BINS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    h[tid] = 0;
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    int64_t kk[RS_ITEMS], pp[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = rs_index(tile0, wv, j, lane);
        kk[j] = i < n ? __builtin_nontemporal_load(key + i) : 0;
        pp[j] = (part != nullptr && i < n) ? __builtin_nontemporal_load(part + i) : pmin;
    }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = rs_index(tile0, wv, j, lane);
        if (i < n) {
            uint64_t c = (uint64_t)kk[j] - (uint64_t)kmin;
            if (part != nullptr) c |= ((uint64_t)pp[j] - (uint64_t)pmin) << kbits;
            if (ibits > 0) c = (c << ibits) | (uint64_t)i;
            comp[i] = c;
            atomicAdd(&h[(c >> shift) & (RS_BINS - 1)], 1u);
        }
    }
    __syncthreads();
    hist[(int64_t)tid * nblocks + blockIdx.x] = h[tid];
}

__global__ __launch_bounds__(RS_BLOCK) void k_rs_hist(const uint64_t* __restrict__ in_key, int64_t n, int shift, uint32_t* __restrict__ hist,
                                                      int64_t nblocks) {
    __shared__ uint32_t h[RS_BINS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    h[tid] = 0;
    __syncthreads();
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    uint64_t k[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = rs_index(tile0, wv, j, lane);
        k[j] = i < n ? __builtin_nontemporal_load(in_key + i) : 0ull;
    }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j)
        if (rs_index(tile0, wv, j, lane) < n) atomicAdd(&h[(k[j] >> shift) & (RS_BINS - 1)], 1u);
    __syncthreads();
    hist[(int64_t)tid * nblocks + blockIdx.x] = h[tid];
}

__device__ __forceinline__ uint32_t bld_wave_excl_scan(uint32_t v) {
    const int lane = lane_id();
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x - v;
}
// exclusive scan over the 256 threads of a workgroup; *total = sum (same in every thread)
__device__ __forceinline__ uint32_t bld_block_excl_scan(uint32_t v, uint32_t* sW /*[RS_WAVES]*/, uint32_t* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ex = bld_wave_excl_scan(v);
    __syncthreads();                                    // sW may still be read from a previous use
    if (lane == 63) sW[wv] = ex + v;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) { const uint32_t s = sW[w]; if (w < wv) base += s; tot += s; }
    *total = tot;
    return base + ex;
}

// one workgroup per digit: hist[d][0..nblocks) becomes the number of elements with digit d in the tiles in front of each tile,
// tot[d] their number over all tiles (the scatter kernel turns the 256 totals into digit bases itself)
__global__ __launch_bounds__(RS_BLOCK) void k_rs_scan(uint32_t* __restrict__ hist, uint32_t* __restrict__ tot, int64_t nblocks) {
    __shared__ uint32_t sW[RS_WAVES];
    const int tid = threadIdx.x, d = blockIdx.x;
    uint32_t* row = hist + (int64_t)d * nblocks;
    const int64_t ipt = (nblocks + RS_BLOCK - 1) / RS_BLOCK;
    const int64_t i0 = (int64_t)tid * ipt, i1 = i0 + ipt < nblocks ? i0 + ipt : nblocks;
    uint32_t local = 0;
    for (int64_t i = i0; i < i1; ++i) local += row[i];
    uint32_t total;
    uint32_t run = bld_block_excl_scan(local, sW, &total);
    for (int64_t i = i0; i < i1; ++i) { const uint32_t c = row[i]; row[i] = run; run += c; }
    if (tid == 0) tot[d] = total;
}

template <bool HAS_VAL>
__global__ __launch_bounds__(RS_BLOCK) void k_rs_scatter(const uint64_t* __restrict__ in_key, const double* __restrict__ in_val, int64_t n, int shift,
                                                         const uint32_t* __restrict__ gbase, const uint32_t* __restrict__ dtot,
                                                         uint64_t* __restrict__ out_key, double* __restrict__ out_val, int64_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
    uint64_t* sKey = reinterpret_cast<uint64_t*>(rs_lds);
    double* sVal = reinterpret_cast<double*>(rs_lds + (size_t)RS_TILE * sizeof(uint64_t));
    __shared__ uint32_t sCnt[RS_WAVES][RS_BINS];
    __shared__ uint32_t sStart[RS_BINS], sBase[RS_BINS], sW[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) sCnt[w][tid] = 0;
    uint64_t k[RS_ITEMS]; double v[RS_ITEMS]; uint32_t r[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = rs_index(tile0, wv, j, lane);
        const bool valid = i < n;
        k[j] = valid ? __builtin_nontemporal_load(in_key + i) : ~0ull;
        v[j] = (HAS_VAL && valid) ? __builtin_nontemporal_load(in_val + i) : 0.0;
    }
    __syncthreads();
    // ranks inside the wave, per digit, in element order: lanes with the same digit find each other with 8 ballots; the lowest
    // of them advances the wave's counter of that digit
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const bool valid = rs_index(tile0, wv, j, lane) < n;
        const uint32_t d = (uint32_t)(k[j] >> shift) & (RS_BINS - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bb = __ballot(bit);
            peers &= bit ? bb : ~bb;
        }
        const int leader = peers ? __ffsll((unsigned long long)peers) - 1 : 0;
        uint32_t old = 0;
        if (valid && lane == leader) { old = sCnt[wv][d]; sCnt[wv][d] = old + (uint32_t)popc64(peers); }
        old = __shfl(old, leader, 64);
        r[j] = old + (uint32_t)popc64(peers & mask_lt(lane));
    }
    __syncthreads();
    {   // thread <-> digit: wave offsets, tile-local start of the digit, its global base
        const uint32_t c0 = sCnt[0][tid], c1 = sCnt[1][tid], c2 = sCnt[2][tid], c3 = sCnt[3][tid];
        sCnt[0][tid] = 0; sCnt[1][tid] = c0; sCnt[2][tid] = c0 + c1; sCnt[3][tid] = c0 + c1 + c2;
        uint32_t tot;
        const uint32_t ex = bld_block_excl_scan(c0 + c1 + c2 + c3, sW, &tot);
        const uint32_t dbase = bld_block_excl_scan(dtot[tid], sW, &tot);       // elements with a smaller digit, all tiles
        sStart[tid] = ex;
        sBase[tid] = dbase + gbase[(int64_t)tid * nblocks + blockIdx.x] - ex;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        if (rs_index(tile0, wv, j, lane) < n) {
            const uint32_t d = (uint32_t)(k[j] >> shift) & (RS_BINS - 1);
            const uint32_t p = sStart[d] + sCnt[wv][d] + r[j];
            sKey[p] = k[j];
            if (HAS_VAL) sVal[p] = v[j];
        }
    }
    __syncthreads();
    const int count = (int)(n - tile0 < RS_TILE ? n - tile0 : RS_TILE);
    for (int i = tid; i < count; i += RS_BLOCK) {
        const uint64_t kk = sKey[i];
        const uint32_t g = sBase[(uint32_t)(kk >> shift) & (RS_BINS - 1)] + (uint32_t)i;
        out_key[g] = kk;
        if (HAS_VAL) out_val[g] = sVal[i];
    }
}

// ---- flags, counts, emit ---------------------------------------------------------------------------------------------
// new-cell / new-partition flags of one element from its predecessor in the sorted order
struct BfFlags { uint64_t fc, fp; };      // ballots over the wave's 64 elements of one iteration
__device__ __forceinline__ BfFlags bf_flags(const uint64_t* __restrict__ comp, int64_t i, int64_t n, uint64_t cur, int kbits, bool has_part, int lane, int ibits) {
    uint64_t prev = __shfl_up(cur, 1, 64);
    if (lane == 0 && i > 0 && i < n) prev = comp[i - 1] >> ibits;
    const bool valid = i < n;
    const bool fc = valid && (i == 0 || cur != prev);
    const bool fp = valid && (i == 0 || (has_part && (cur >> kbits) != (prev >> kbits)));
    BfFlags f; f.fc = __ballot(fc); f.fp = __ballot(fp);
    return f;
}

__global__ __launch_bounds__(RS_BLOCK) void k_bf_count(const uint64_t* __restrict__ comp, int64_t n, int kbits, int ibits, int has_part,
                                                       uint32_t* __restrict__ cnt_c, uint32_t* __restrict__ cnt_p) {
    __shared__ uint32_t sC[RS_WAVES], sP[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    uint32_t c = 0, p = 0;
#pragma unroll 4
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = rs_index(tile0, wv, j, lane);
        const uint64_t cur = i < n ? comp[i] >> ibits : 0ull;
        const BfFlags f = bf_flags(comp, i, n, cur, kbits, has_part != 0, lane, ibits);
        c += (uint32_t)popc64(f.fc); p += (uint32_t)popc64(f.fp);
    }
    if (lane == 0) { sC[wv] = c; sP[wv] = p; }
    __syncthreads();
    if (tid == 0) { cnt_c[blockIdx.x] = sC[0] + sC[1] + sC[2] + sC[3]; cnt_p[blockIdx.x] = sP[0] + sP[1] + sP[2] + sP[3]; }
}

// one workgroup: exclusive prefixes of the per-tile counts (in place), totals to the control block
// host (pinned, may be null): the two totals and then the word [5] = seq straight into the host's control block — the caller polls for it
// instead of a copy command + a stream wait (2 x ~40 us of host latency in a 1.5 ms build of both orientations)
__global__ __launch_bounds__(1024) void k_bf_scan(uint32_t* __restrict__ cnt_c, uint32_t* __restrict__ cnt_p, int64_t ntiles, BuildCtl* ctl,
                                                  unsigned long long* host = nullptr, unsigned long long seq = 0) {
    __shared__ uint32_t wsum[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t carry[2] = {0, 0};
    uint32_t* arr[2] = {cnt_c, cnt_p};
    for (int64_t base = 0; base < ntiles; base += 1024) {
        const int64_t i = base + tid;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const uint32_t v = i < ntiles ? arr[a][i] : 0u;
            const uint32_t ex = bld_wave_excl_scan(v);
            if (lane == 63) wsum[a][wv] = ex + v;
            __syncthreads();
            uint32_t run = carry[a] + ex, total = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { const uint32_t s = wsum[a][k]; if (k < wv) run += s; total += s; }
            if (i < ntiles) arr[a][i] = run;
            carry[a] += total;
            __syncthreads();
        }
    }
    if (tid == 0) {
        ctl->ncells = carry[0]; ctl->nparts = carry[1]; ctl->nlong = 0;
        if (host != nullptr) {
            __hip_atomic_store(host + 4, (unsigned long long)carry[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // BuildCtl::ncells
            __hip_atomic_store(host + 5, (unsigned long long)carry[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // BuildCtl::nparts
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(host + 7, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);                               // BuildCtl::zeros doubles as the sequence word here
        }
    }
}
// waits for k_bf_scan's hand-over (polling the pinned word; the stream is asked now and then so that a failed launch cannot hang the host)
static hipError_t wait_counts(BuildCtl* hctl, unsigned long long seq, hipStream_t stream) {
    volatile unsigned long long* w = reinterpret_cast<volatile unsigned long long*>(hctl) + 7;
    auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    while (__atomic_load_n(w, __ATOMIC_ACQUIRE) != seq) {
        if (std::chrono::steady_clock::now() < next_query) continue;
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipErrorNotReady) { next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2); continue; }
        if (q != hipSuccess) return q;
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) != seq) return hipErrorUnknown;
    }
    return hipSuccess;
}
static std::atomic<unsigned long long> g_count_seq{0x5eed0000ull};

__device__ __forceinline__ double bld_combine(double a, double b, int32_t combine) {
    return combine == 0 ? a + b : (combine == 1 ? a * b : b);
}

// mode 0: mapped partitions (ids = rank of the distinct partition keys, semaphore emitted by the first cell of a partition)
// mode 1: plain vector (no semaphores)   mode 2: explicit partition ids 1..P (semaphores by k_emit_sems)
// ibits > 0: the words are composite << ibits | input index and `val` is the caller's value array (gathered by index)
__global__ __launch_bounds__(RS_BLOCK) void k_bf_emit(const uint64_t* __restrict__ comp, const double* __restrict__ val, int64_t n, int kbits, int ibits,
                                                      int64_t pmin, int64_t kmin, int mode, int32_t combine,
                                                      const uint32_t* __restrict__ off_c, const uint32_t* __restrict__ off_p,
                                                      KeyArr out_keys, double* __restrict__ out_vals, int64_t* __restrict__ part_keys,
                                                      uint32_t* __restrict__ scell, BuildCtl* ctl, LongRun* __restrict__ queue,
                                                      uint64_t* __restrict__ der_comp, double* __restrict__ der_val, int pbits) {
    // der_comp / der_val (or nullptr): the folded cells once more, in this orientation's order, as the composites of the TWIN orientation
    // — (key - kmin) << pbits | (partition - pmin) — with their values: what the twin's builder sorts by its own partition bits
    // alone instead of starting from the caller's triples again (mat_build_both_dev)
    __shared__ uint32_t sC[RS_WAVES], sP[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    const bool has_part = mode != 1;
    const uint64_t kmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
    const uint64_t imask = ibits > 0 ? ((1ull << ibits) - 1ull) : 0ull;
    uint64_t cur[RS_ITEMS], fcb[RS_ITEMS], fpb[RS_ITEMS];
    double vj[RS_ITEMS];
    uint32_t c = 0, p = 0;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = rs_index(tile0, wv, j, lane);
        const uint64_t w = i < n ? __builtin_nontemporal_load(comp + i) : 0ull;
        cur[j] = w >> ibits;
        vj[j] = i < n ? (ibits > 0 ? val[w & imask] : __builtin_nontemporal_load(val + i)) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = rs_index(tile0, wv, j, lane);
        const BfFlags f = bf_flags(comp, i, n, cur[j], kbits, has_part, lane, ibits);
        fcb[j] = f.fc; fpb[j] = f.fp;
        c += (uint32_t)popc64(f.fc); p += (uint32_t)popc64(f.fp);
    }
    if (lane == 0) { sC[wv] = c; sP[wv] = p; }
    __syncthreads();
    uint32_t rc = off_c[blockIdx.x], rp = off_p[blockIdx.x];       // cells / partition starts in front of this wave's elements
    for (int w = 0; w < wv; ++w) { rc += sC[w]; rp += sP[w]; }
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = rs_index(tile0, wv, j, lane);
        const uint64_t le = lane == 63 ? ~0ull : mask_lt(lane + 1);
        const uint32_t rank = rc + (uint32_t)popc64(fcb[j] & le);            // 1-based rank of the cell this element belongs to
        const uint32_t pidr = rp + (uint32_t)popc64(fpb[j] & le);            // 1-based rank of its partition
        uint64_t nxt = __shfl_down(cur[j], 1, 64);
        if (lane == 63) nxt = i + 1 < n ? comp[i + 1] >> ibits : ~cur[j];
        if (i + 1 >= n) nxt = ~cur[j];
        if (i < n) {
            if (scell != nullptr) scell[i] = rank;
            if ((fcb[j] >> lane) & 1ull) {
                const uint64_t cc = cur[j];
                const int64_t key = (int64_t)((cc & kmask) + (uint64_t)kmin);
                const int64_t part = has_part ? (int64_t)((kbits >= 64 ? 0ull : (cc >> kbits)) + (uint64_t)pmin) : 0;
                const int64_t pid = mode == 0 ? (int64_t)pidr : (mode == 1 ? 0 : part);
                const int64_t pos = (int64_t)rank - 1 + pid;
                double acc = vj[j];
                if (nxt == cc) {                                  // duplicates: left fold in input order  src/pcsr.jl:374-375
                    int64_t t = i + 1;
                    int len = 1;
                    while (t < n && len < FOLD_INLINE && (comp[t] >> ibits) == cc) {
                        acc = bld_combine(acc, ibits > 0 ? val[comp[t] & imask] : val[t], combine); ++t; ++len;
                    }
                    if (t < n && len == FOLD_INLINE && (comp[t] >> ibits) == cc) {      // a long run: a wave finishes it (k_fold_long)
                        const unsigned long long q = atomicAdd(&ctl->nlong, 1ull);
                        LongRun lr; lr.next = t; lr.pos = pos; lr.acc = acc; lr.comp = cc; lr.dpos = (int64_t)rank - 1;
                        queue[q] = lr;
                    }
                }
                out_keys[pos] = key;
                out_vals[pos] = acc;
                if (der_comp != nullptr) {
                    der_comp[rank - 1] = ((cc & kmask) << pbits) | (kbits >= 64 ? 0ull : (cc >> kbits));
                    der_val[rank - 1] = acc;
                }
                if (mode == 0 && ((fpb[j] >> lane) & 1ull)) {
                    out_keys[pos - 1] = SEM_KEY;
                    out_vals[pos - 1] = (double)pid;
                    part_keys[pid - 1] = part;
                }
            }
        }
        rc += (uint32_t)popc64(fcb[j]); rp += (uint32_t)popc64(fpb[j]);
    }
}

// the rest of the duplicate runs longer than FOLD_INLINE: one wave per run, 64 values per coalesced load, folded in order
__global__ __launch_bounds__(64) void k_fold_long(const uint64_t* __restrict__ comp, const double* __restrict__ val, int64_t n, int ibits, int32_t combine,
                                                  const BuildCtl* ctl, const LongRun* __restrict__ queue, double* __restrict__ out_vals,
                                                  double* __restrict__ der_val) {
    const uint64_t imask = ibits > 0 ? ((1ull << ibits) - 1ull) : 0ull;
    const int lane = threadIdx.x;
    const unsigned long long nl = ctl->nlong;
    for (unsigned long long q = blockIdx.x; q < nl; q += gridDim.x) {
        const LongRun lr = queue[q];
        double acc = lr.acc;
        int64_t t = lr.next;
        while (true) {
            const int64_t i = t + lane;
            const uint64_t w = i < n ? comp[i] : 0ull;
            const bool same = i < n && (w >> ibits) == lr.comp;
            const double v = same ? (ibits > 0 ? val[w & imask] : val[i]) : 0.0;
            const uint64_t b = __ballot(same);
            const int cnt = b == ~0ull ? 64 : __ffsll((unsigned long long)~b) - 1;      // leading run of equal composites
            for (int l = 0; l < cnt; ++l) acc = bld_combine(acc, __shfl(v, l, 64), combine);
            if (cnt < 64) break;
            t += 64;
        }
        if (lane == 0) { out_vals[lr.pos] = acc; if (der_val != nullptr) der_val[lr.dpos] = acc; }
    }
}

// mode 2: semaphore cell of every partition p = 1..P (empty partitions included, src/pcsr.jl:36-41):
// position = (#distinct cells of partitions < p) + p - 1
__global__ void k_emit_sems(const uint64_t* __restrict__ comp, int kbits, int ibits, int64_t pmin, const uint32_t* __restrict__ scell, int64_t n,
                            int64_t nparts, KeyArr out_keys, double* __restrict__ out_vals) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x + 1;
    if (p > nparts) return;
    int64_t lo = 0, hi = n;                     // first index with part >= p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t pm = (int64_t)((kbits >= 64 ? 0ull : ((comp[mid] >> ibits) >> kbits)) + (uint64_t)pmin);
        if (pm < p) lo = mid + 1; else hi = mid;
    }
    const int64_t cells_before = lo == 0 ? 0 : (int64_t)scell[lo - 1];
    const int64_t pos = cells_before + p - 1;
    out_keys[pos] = SEM_KEY;
    out_vals[pos] = (double)p;
}

// ---- general path (composite wider than 64 bits): TWO runs of the stable radix sort above, each over 64-bit digits' worth of one
// component, carrying the input index as the 8-byte payload — by key first, then (stable) by partition: (partition, key, input order),
// the order the single composite sort produces.  (Round 3 called a library sort here.)
__global__ void k_wide_key0(const int64_t* __restrict__ key, int64_t kmin, uint64_t* __restrict__ comp, double* __restrict__ payload, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        comp[i] = (uint64_t)key[i] - (uint64_t)kmin;                 // order-preserving: the range fits 64 bits
        payload[i] = __longlong_as_double((long long)i);              // (moved as raw bits, never computed on)
    }
}
// the composites of the second run, in the order the first one left: partition of the element behind each payload
__global__ void k_wide_part(const int64_t* __restrict__ part, int64_t pmin, const double* __restrict__ payload, uint64_t* __restrict__ comp, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        comp[j] = (uint64_t)part[__double_as_longlong(payload[j])] - (uint64_t)pmin;
}
// sorted order -> input index, key and partition of every element
__global__ void k_wide_finish(const int64_t* __restrict__ part, const int64_t* __restrict__ key, const double* __restrict__ payload,
                              uint32_t* __restrict__ idx, int64_t* __restrict__ k2, int64_t* __restrict__ p2, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = __double_as_longlong(payload[j]);
        idx[j] = (uint32_t)i;
        k2[j] = key[i];
        if (part != nullptr) p2[j] = part[i];
    }
}
// inclusive prefix sums of the two flag arrays: per-tile sums (k_bf_scan turns them into tile offsets and totals), then the tiles
__global__ __launch_bounds__(RS_BLOCK) void k_flag_tile_sums(const uint32_t* __restrict__ fpart, const uint32_t* __restrict__ fcell, int64_t n,
                                                             uint32_t* __restrict__ cnt_c, uint32_t* __restrict__ cnt_p) {
    __shared__ uint32_t sC[RS_WAVES], sP[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    uint32_t c = 0, p = 0;
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int64_t i = tile0 + (int64_t)j * RS_BLOCK + tid;
        if (i < n) { c += fcell[i]; p += fpart[i]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); p += __shfl_xor(p, o, 64); }
    if (lane == 0) { sC[wv] = c; sP[wv] = p; }
    __syncthreads();
    if (tid == 0) { cnt_c[blockIdx.x] = sC[0] + sC[1] + sC[2] + sC[3]; cnt_p[blockIdx.x] = sP[0] + sP[1] + sP[2] + sP[3]; }
}
__global__ __launch_bounds__(RS_BLOCK) void k_flag_scan_apply(const uint32_t* __restrict__ fpart, const uint32_t* __restrict__ fcell, int64_t n,
                                                              const uint32_t* __restrict__ off_c, const uint32_t* __restrict__ off_p,
                                                              uint32_t* __restrict__ spart, uint32_t* __restrict__ scell) {
    __shared__ uint32_t sW[RS_WAVES];
    const int tid = threadIdx.x;
    const int64_t tile0 = (int64_t)blockIdx.x * RS_TILE;
    uint32_t run_c = off_c[blockIdx.x], run_p = off_p[blockIdx.x];
    for (int j = 0; j < RS_ITEMS; ++j) {                       // RS_BLOCK consecutive elements per step
        const int64_t i = tile0 + (int64_t)j * RS_BLOCK + tid;
        const uint32_t fc = i < n ? fcell[i] : 0u, fp = i < n ? fpart[i] : 0u;
        uint32_t tc, tp;
        const uint32_t ec = bld_block_excl_scan(fc, sW, &tc);
        const uint32_t ep = bld_block_excl_scan(fp, sW, &tp);
        if (i < n) { scell[i] = run_c + ec + fc; spart[i] = run_p + ep + fp; }
        run_c += tc; run_p += tp;
    }
}
__global__ void k_flags(const int64_t* __restrict__ part, const int64_t* __restrict__ key, uint32_t* __restrict__ fpart,
                        uint32_t* __restrict__ fcell, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool np = part == nullptr ? (i == 0) : ((i == 0) || part[i] != part[i - 1]);
    fpart[i] = np ? 1u : 0u;
    fcell[i] = (np || key[i] != key[i - 1]) ? 1u : 0u;
}
__global__ void k_emit_wide(const int64_t* __restrict__ part, const int64_t* __restrict__ key, const uint32_t* __restrict__ idx,
                            const double* __restrict__ val, const uint32_t* __restrict__ fpart, const uint32_t* __restrict__ fcell,
                            const uint32_t* __restrict__ spart, const uint32_t* __restrict__ scell, int64_t n, int32_t combine,
                            KeyArr out_keys, double* __restrict__ out_vals, int64_t* __restrict__ part_keys, int mode) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!fcell[i]) return;
    const int64_t pid = mode == 0 ? (int64_t)spart[i] : (mode == 1 ? 0 : part[i]);
    const int64_t rank = scell[i];
    double acc = val[idx[i]];
    for (int64_t j = i + 1; j < n && !fcell[j]; ++j) acc = bld_combine(acc, val[idx[j]], combine);
    const int64_t pos = rank - 1 + pid;
    out_keys[pos] = key[i];
    out_vals[pos] = acc;
    if (mode == 0 && fpart[i]) {
        out_keys[pos - 1] = SEM_KEY;
        out_vals[pos - 1] = (double)pid;
        part_keys[pid - 1] = part[i];
    }
}
__global__ void k_emit_sems_wide(const int64_t* __restrict__ part_sorted, const uint32_t* __restrict__ scell, int64_t n, int64_t nparts,
                                 KeyArr out_keys, double* __restrict__ out_vals) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x + 1;
    if (p > nparts) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (part_sorted[mid] < p) lo = mid + 1; else hi = mid;
    }
    const int64_t cells_before = lo == 0 ? 0 : (int64_t)scell[lo - 1];
    const int64_t pos = cells_before + p - 1;
    out_keys[pos] = SEM_KEY;
    out_vals[pos] = (double)p;
}

// ---- host side ----------------------------------------------------------------------------------------------------------
// The scratch of a build is ONE block from the caching allocator (pool.hip), carved into its arrays: a second build of the same
// size finds it again without a driver call.
// the pinned mirror of BuildCtl: a handful of them are kept (hipHostMalloc / hipHostFree cost more than a sort pass)
static std::mutex g_pin_mu;
static std::vector<void*> g_pin_free;
static hipError_t pinned_ctl_get(void** out) {
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (!g_pin_free.empty()) { *out = g_pin_free.back(); g_pin_free.pop_back(); return hipSuccess; }
    }
    return hipHostMalloc(out, 256, hipHostMallocDefault);
}
static void pinned_ctl_put(void* p) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_free.size() < 16) g_pin_free.push_back(p); else (void)hipHostFree(p);
}
static void free_scratch(BuildScratch& s) {
    if (s.base) pool_free(s.base);
    if (s.base_val) pool_free(s.base_val);
    if (s.h_ctl) pinned_ctl_put(s.h_ctl);
    s = BuildScratch();
}

// error path: the scratch comes from the caching allocator (pool.hip), which hands a freed block out again at once — earlier passes
// may still be in flight on it, so the build stream is drained first (build_abort does the same)
#define BCHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { if (s.stream) (void)hipStreamSynchronize(s.stream); free_scratch(s); return _e; } } while (0)

static int bit_width_u64(uint64_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; }

static hipError_t prepare_wide(const int64_t* d_part, const int64_t* d_key, int64_t nnz, BuildScratch& s, int64_t counts[2], hipStream_t stream,
                               int64_t kmin, int kbits, int64_t pmin, int pbits);

// value ranges and reserved-key check of two key arrays that are already in HBM (a and b: rows and columns of a triple stream): one
// pass on the device instead of a host loop over arrays the host would otherwise not touch at all; synchronises the stream
// ---- a vector of up to 1024 entries in ONE launch (dynamicsparsevec of a small input, src/vector.jl:10-62): the workgroup reads the caller's
// (key, value) pairs from a pinned landing area, ranks them (stable: equal keys keep their input order), folds equal keys left to right
// with `combine` (_prepare_keys_vals!, src/vector.jl:10-36) and writes the packed stream in front of the slot buffers; the number of
// entries goes back through pinned memory.  The general builder needs ~12 launches and a read-back for the same 50 entries (130 us).
// io: [0] = m (out), [5] = sequence number (out), [8, 8 + cap) keys, [8 + cap, 8 + 2 cap) values (in)
constexpr int BSV_MAX = 1024;
__global__ __launch_bounds__(BSV_MAX) void k_build_small_vec(int64_t* io, int n, int cap, int32_t combine, KeyArr out_k, double* __restrict__ out_v,
                                                             unsigned long long seq) {
    __shared__ int64_t sk[BSV_MAX], tk[BSV_MAX];
    __shared__ double sv[BSV_MAX], tv[BSV_MAX];
    __shared__ int sWave[BSV_MAX / 64];
    const int i = threadIdx.x, lane = i & 63, wv = i >> 6;
    int64_t key = 0; double val = 0.0;
    if (i < n) {
        key = __hip_atomic_load(io + 8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        val = __longlong_as_double(__hip_atomic_load(io + 8 + cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        sk[i] = key; sv[i] = val;
    }
    __syncthreads();
    if (i < n) {
        int r = 0;
        for (int j = 0; j < n; ++j) { const int64_t kj = sk[j]; r += (kj < key || (kj == key && j < i)) ? 1 : 0; }
        tk[r] = key; tv[r] = val;
    }
    __syncthreads();
    // heads of the runs of equal keys, their output positions, the left fold of each run
    const bool head = i < n && (i == 0 || tk[i] != tk[i - 1]);
    const uint64_t hb = __ballot(head);
    if (lane == 0) sWave[wv] = __popcll(hb);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < BSV_MAX / 64; ++w) { const int c = sWave[w]; if (w < wv) before += c; total += c; }
    if (head) {
        const int pos = before + __popcll(hb & mask_lt(lane));
        const int64_t k0 = tk[i];
        double acc = tv[i];
        for (int t = i + 1; t < n && tk[t] == k0; ++t) acc = bld_combine(acc, tv[t], combine);
        out_k[pos] = k0; out_v[pos] = acc;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (i == 0) {
        __hip_atomic_store(io, (int64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(io) + 5, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
hipError_t launch_build_small_vec(int64_t* io, int n, int cap, int32_t combine, KeyArr out_k, double* out_v, unsigned long long seq, hipStream_t stream) {
    if (n < 1 || n > BSV_MAX || cap < n) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_build_small_vec, dim3(1), dim3(BSV_MAX), 0, stream, io, n, cap, combine, out_k, out_v, seq);
    return hipGetLastError();
}

hipError_t device_key_scan(const int64_t* d_a, const int64_t* d_b, int64_t n, KeyRange* ra, KeyRange* rb, bool* a_zero, bool* b_zero,
                           hipStream_t stream) {
    *ra = KeyRange(); *rb = KeyRange(); *a_zero = *b_zero = false;
    if (n <= 0) return hipSuccess;
    void* base = nullptr; void* pin = nullptr;
    hipError_t e = pool_alloc(&base, 256 + 4 * MM_BLOCKS * 8 + 64);
    if (e != hipSuccess) return e;
    e = pinned_ctl_get(&pin);
    if (e != hipSuccess) { pool_free(base); return e; }
    BuildCtl* dctl = static_cast<BuildCtl*>(base);
    long long* partial = reinterpret_cast<long long*>(static_cast<char*>(base) + 256);
    unsigned int* ticket = reinterpret_cast<unsigned int*>(partial + 4 * MM_BLOCKS);
    e = hipMemsetAsync(ticket, 0, 2 * sizeof(unsigned int), stream);
    if (e == hipSuccess) {
        const int mm_blocks = (int)std::min<int64_t>(MM_BLOCKS, (n + 1023) / 1024);
        hipLaunchKernelGGL(k_minmax, dim3(mm_blocks), dim3(256), 0, stream, d_a, d_b, n, dctl, partial, ticket);
        e = hipMemcpyAsync(pin, dctl, sizeof(BuildCtl), hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) {
        const BuildCtl* h = static_cast<const BuildCtl*>(pin);
        ra->lo = h->pmin; ra->hi = h->pmax; rb->lo = h->kmin; rb->hi = h->kmax;
        *a_zero = (h->zeros & 1ull) != 0; *b_zero = (h->zeros & 2ull) != 0;
    }
    if (e != hipSuccess) (void)hipStreamSynchronize(stream);      // a failed copy / wait: the kernel may still be running on the block
    pinned_ctl_put(pin);
    pool_free(base);
    return e;
}

// Phase 1: sort + flags + counts.  d_part / d_key / d_val: the nnz input triples in HBM (d_part == nullptr: a plain vector).
// Returns the number of distinct cells and of partitions through counts[0..1] (host).  The scratch stays alive for phase 2
// (build_emit), which writes counts[0] + #partitions stream cells.
hipError_t build_prepare(const int64_t* d_part, const int64_t* d_key, const double* d_val, int64_t nnz, KeyRange part_range, KeyRange key_range,
                         BuildScratch& s, int64_t counts[2], hipStream_t stream, const std::function<void()>* while_sorting) {
    s = BuildScratch();
    s.n = nnz; s.stream = stream;
    const size_t n = (size_t)nnz;
    static const bool force_wide = [] { const char* e = dev_env("DSA_BUILD_WIDE"); return e && e[0] == '1'; }();
    static const bool force_minmax = [] { const char* e = dev_env("DSA_BUILD_MINMAX"); return e && e[0] == '1'; }();      // dev: ignore the caller's ranges
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int64_t nblocks = (nnz + RS_TILE - 1) / RS_TILE;
    BCHK(pinned_ctl_get(&s.h_ctl));
    const size_t b_ctl = up(sizeof(BuildCtl)), b_gh = up(8 * RS_BINS * 4 + 4 * MM_BLOCKS * 8 + 64), b_comp = up(n * 8),
                 b_hist = up((size_t)RS_BINS * nblocks * 4), b_cnt = up((size_t)(nblocks + 1) * 4),
                 b_queue = up((n / FOLD_INLINE + 2) * sizeof(LongRun)), b_scell = up(n * 4);
    // (the two value arrays — 16 bytes per triple — are a block of their own, taken further down only when the values really travel
    // with the sorted words: the index-in-word sort of config 3 never touches them, and the block used to be held, untouched, all the same)
    BCHK(pool_alloc(&s.base, b_ctl + b_gh + 2 * b_comp + b_hist + 2 * b_cnt + b_queue + b_scell));
    char* q = static_cast<char*>(s.base);
    auto take = [&q](size_t b) { char* r = q; q += b; return r; };
    s.d_ctl = take(b_ctl); s.ghist = (uint32_t*)take(b_gh);
    s.comp[0] = (uint64_t*)take(b_comp); s.comp[1] = (uint64_t*)take(b_comp);
    s.hist = (uint32_t*)take(b_hist); s.cnt_c = (uint32_t*)take(b_cnt); s.cnt_p = (uint32_t*)take(b_cnt);
    s.queue = take(b_queue); s.scell = (uint32_t*)take(b_scell);
    BuildCtl* dctl = static_cast<BuildCtl*>(s.d_ctl);
    BuildCtl* hctl = static_cast<BuildCtl*>(s.h_ctl);
    // ---- key ranges: how many bits the composite needs
    if (!key_range.known() || (d_part != nullptr && !part_range.known()) || force_minmax) {
        long long* partial = reinterpret_cast<long long*>(s.ghist + 8 * RS_BINS);
        unsigned int* ticket = reinterpret_cast<unsigned int*>(partial + 4 * MM_BLOCKS);
        BCHK(hipMemsetAsync(ticket, 0, 2 * sizeof(unsigned int), stream));
        const int mm_blocks = (int)std::min<int64_t>(MM_BLOCKS, (nnz + 1023) / 1024);
        hipLaunchKernelGGL(k_minmax, dim3(mm_blocks), dim3(256), 0, stream, d_part, d_key, nnz, dctl, partial, ticket);
        BCHK(hipMemcpyAsync(hctl, dctl, sizeof(BuildCtl), hipMemcpyDeviceToHost, stream));
        BCHK(hipStreamSynchronize(stream));
        key_range.lo = hctl->kmin; key_range.hi = hctl->kmax;
        if (d_part) { part_range.lo = hctl->pmin; part_range.hi = hctl->pmax; }
    }
    s.kmin = key_range.lo; s.pmin = d_part ? part_range.lo : 0;
    s.kbits = std::max(1, bit_width_u64((uint64_t)key_range.hi - (uint64_t)key_range.lo));
    s.pbits = d_part ? bit_width_u64((uint64_t)part_range.hi - (uint64_t)part_range.lo) : 0;
    if (s.kbits + s.pbits > 64 || force_wide) {
        const int64_t kmin = s.kmin, pmin = s.pmin;
        const int kbits = s.kbits, pbits = s.pbits;
        free_scratch(s);
        s.n = nnz; s.stream = stream; s.wide_path = true;
        return prepare_wide(d_part, d_key, nnz, s, counts, stream, kmin, kbits, pmin, pbits);
    }
    // ---- the sort: passes over bits [0, kbits + pbits).  When composite + input index fit one 64-bit word the passes move 8-byte
    //      words without a payload (half the bytes, half the LDS per tile) and the values are gathered by index at the emit
    static PerDeviceOnce once;
    const size_t lds_bytes = (size_t)RS_TILE * (sizeof(uint64_t) + sizeof(double));
    BCHK(once.run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(k_rs_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }));
    const int total_bits = s.kbits + s.pbits;
    static const bool idx_sort = [] { const char* e = dev_env("DSA_BUILD_IDXSORT"); return !(e && e[0] == '0'); }();      // dev knob: 0 = always carry the values
    const int ibits_need = std::max(1, bit_width_u64((uint64_t)(nnz - 1)));
    s.ibits = (idx_sort && total_bits + ibits_need <= 64) ? ibits_need : 0;
    if (s.ibits == 0) {
        BCHK(pool_alloc(&s.base_val, 2 * b_comp));
        s.val[0] = static_cast<double*>(s.base_val); s.val[1] = reinterpret_cast<double*>(static_cast<char*>(s.base_val) + b_comp);
    }
    const int npass = (total_bits + 7) / 8;
    const dim3 grid((unsigned)nblocks), block(RS_BLOCK);
    int cur = 0;                                    // comp[cur] holds the current order; values: d_val before the first scatter
    const double* vin = d_val;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = s.ibits + 8 * pass;
        uint32_t* dtot = s.ghist + pass * RS_BINS;
        if (pass == 0) hipLaunchKernelGGL(k_comp_hist, grid, block, 0, stream, s.pbits > 0 ? d_part : (const int64_t*)nullptr, d_key, nnz, s.pmin, s.kmin, s.kbits, s.ibits, s.comp[0], shift, s.hist, nblocks);
        else hipLaunchKernelGGL(k_rs_hist, grid, block, 0, stream, (const uint64_t*)s.comp[cur], nnz, shift, s.hist, nblocks);
        hipLaunchKernelGGL(k_rs_scan, dim3(RS_BINS), block, 0, stream, s.hist, dtot, nblocks);
        if (s.ibits > 0)
            hipLaunchKernelGGL(k_rs_scatter<false>, grid, block, (size_t)RS_TILE * sizeof(uint64_t), stream, (const uint64_t*)s.comp[cur], (const double*)nullptr, nnz, shift,
                               (const uint32_t*)s.hist, (const uint32_t*)dtot, s.comp[1 - cur], (double*)nullptr, nblocks);
        else
            hipLaunchKernelGGL(k_rs_scatter<true>, grid, block, lds_bytes, stream, (const uint64_t*)s.comp[cur], vin, nnz, shift, (const uint32_t*)s.hist,
                               (const uint32_t*)dtot, s.comp[1 - cur], s.val[1 - cur], nblocks);
        cur = 1 - cur;
        if (s.ibits == 0) vin = s.val[cur];
    }
    s.sorted = cur;
    s.vsorted = vin;                                // (ibits > 0: still the caller's array)
    // ---- flags: per-tile counts of new cells / new partitions, their prefixes, the totals
    hipLaunchKernelGGL(k_bf_count, grid, block, 0, stream, (const uint64_t*)s.comp[cur], nnz, s.kbits, s.ibits, d_part ? 1 : 0, s.cnt_c, s.cnt_p);
    const unsigned long long seq = ++g_count_seq;
    hctl->zeros = 0;
    hipLaunchKernelGGL(k_bf_scan, dim3(1), dim3(1024), 0, stream, s.cnt_c, s.cnt_p, nblocks, dctl, reinterpret_cast<unsigned long long*>(hctl), seq);
    // everything above is in flight: host work of the caller that does not need the counts (allocations) goes here
    if (while_sorting && *while_sorting) {
        try { (*while_sorting)(); } catch (...) { (void)hipStreamSynchronize(stream); throw; }
    }
    BCHK(wait_counts(hctl, seq, stream));
    counts[0] = (int64_t)hctl->ncells; counts[1] = (int64_t)hctl->nparts;
    return hipGetLastError();
}

// ---- the TWIN orientation of a matrix from the cells its sibling has just emitted (mat_build_both_dev) ------------------------------
// The sibling's emit leaves the n folded cells as composites of THIS orientation — partition bits above the key bits — in the
// sibling's order: ascending by (key, partition).  A stable sort by the partition bits alone (ceil(pbits / 8) passes instead of
// ceil((pbits + kbits) / 8), 16-byte records: word + value) puts them into (partition, key) order; nothing is folded (the cells are
// distinct), no composite pass, no gather of the values at the emit.  build_derived_alloc: the scratch, whose comp[0] / val[0] the
// sibling's emit fills; build_derived_sort: passes, flags, counts (one stream wait); then build_emit as for any other build.
hipError_t build_derived_alloc(BuildScratch& s, int64_t n, int kbits, int pbits, int64_t kmin, int64_t pmin, hipStream_t stream) {
    s = BuildScratch();
    s.n = n; s.stream = stream;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int64_t nblocks = (n + RS_TILE - 1) / RS_TILE;
    BCHK(pinned_ctl_get(&s.h_ctl));
    const size_t nn = (size_t)n;
    const size_t b_ctl = up(sizeof(BuildCtl)), b_gh = up(8 * RS_BINS * 4 + 4 * MM_BLOCKS * 8 + 64), b_comp = up(nn * 8),
                 b_hist = up((size_t)RS_BINS * nblocks * 4), b_cnt = up((size_t)(nblocks + 1) * 4), b_queue = up((nn / FOLD_INLINE + 2) * sizeof(LongRun));
    BCHK(pool_alloc(&s.base, b_ctl + b_gh + 2 * b_comp + b_hist + 2 * b_cnt + b_queue));
    char* q = static_cast<char*>(s.base);
    auto take = [&q](size_t b) { char* r = q; q += b; return r; };
    s.d_ctl = take(b_ctl); s.ghist = (uint32_t*)take(b_gh);
    s.comp[0] = (uint64_t*)take(b_comp); s.comp[1] = (uint64_t*)take(b_comp);
    s.hist = (uint32_t*)take(b_hist); s.cnt_c = (uint32_t*)take(b_cnt); s.cnt_p = (uint32_t*)take(b_cnt);
    s.queue = take(b_queue); s.scell = nullptr;
    BCHK(pool_alloc(&s.base_val, 2 * b_comp));
    s.val[0] = static_cast<double*>(s.base_val); s.val[1] = reinterpret_cast<double*>(static_cast<char*>(s.base_val) + b_comp);
    s.kmin = kmin; s.pmin = pmin; s.kbits = kbits; s.pbits = pbits; s.ibits = 0;
    return hipSuccess;
}
hipError_t build_derived_sort(BuildScratch& s, int64_t counts[2], hipStream_t stream, const std::function<void()>* while_sorting) {
    const int64_t n = s.n;
    const int64_t nblocks = (n + RS_TILE - 1) / RS_TILE;
    BuildCtl* dctl = static_cast<BuildCtl*>(s.d_ctl);
    BuildCtl* hctl = static_cast<BuildCtl*>(s.h_ctl);
    static PerDeviceOnce once;
    const size_t lds_bytes = (size_t)RS_TILE * (sizeof(uint64_t) + sizeof(double));
    BCHK(once.run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(k_rs_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }));
    const int npass = (s.pbits + 7) / 8;
    const dim3 grid((unsigned)nblocks), block(RS_BLOCK);
    int cur = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = s.kbits + 8 * pass;
        uint32_t* dtot = s.ghist + pass * RS_BINS;
        hipLaunchKernelGGL(k_rs_hist, grid, block, 0, stream, (const uint64_t*)s.comp[cur], n, shift, s.hist, nblocks);
        hipLaunchKernelGGL(k_rs_scan, dim3(RS_BINS), block, 0, stream, s.hist, dtot, nblocks);
        hipLaunchKernelGGL(k_rs_scatter<true>, grid, block, lds_bytes, stream, (const uint64_t*)s.comp[cur], (const double*)s.val[cur], n, shift, (const uint32_t*)s.hist,
                           (const uint32_t*)dtot, s.comp[1 - cur], s.val[1 - cur], nblocks);
        cur = 1 - cur;
    }
    s.sorted = cur;
    s.vsorted = s.val[cur];
    hipLaunchKernelGGL(k_bf_count, grid, block, 0, stream, (const uint64_t*)s.comp[cur], n, s.kbits, 0, 1, s.cnt_c, s.cnt_p);
    const unsigned long long seq = ++g_count_seq;
    hctl->zeros = 0;
    hipLaunchKernelGGL(k_bf_scan, dim3(1), dim3(1024), 0, stream, s.cnt_c, s.cnt_p, nblocks, dctl, reinterpret_cast<unsigned long long*>(hctl), seq);
    if (while_sorting && *while_sorting) {
        try { (*while_sorting)(); } catch (...) { (void)hipStreamSynchronize(stream); throw; }
    }
    BCHK(wait_counts(hctl, seq, stream));
    counts[0] = (int64_t)hctl->ncells; counts[1] = (int64_t)hctl->nparts;
    return hipGetLastError();
}

static hipError_t emit_wide(const double* d_val, int32_t combine, BuildScratch& s, KeyArr out_keys, double* out_vals, int64_t* part_keys, int mode,
                            int64_t nparts_explicit, hipStream_t stream, bool wait_and_free);

hipError_t build_emit(const double* d_val, int32_t combine, BuildScratch& s, KeyArr out_keys, double* out_vals,
                      int64_t* part_keys, int mode, int64_t nparts_explicit, hipStream_t stream, bool wait_and_free,
                      uint64_t* der_comp, double* der_val) {
    if (s.wide_path) return emit_wide(d_val, combine, s, out_keys, out_vals, part_keys, mode, nparts_explicit, stream, wait_and_free);
    const int64_t nblocks = (s.n + RS_TILE - 1) / RS_TILE;
    BuildCtl* dctl = static_cast<BuildCtl*>(s.d_ctl);
    const uint64_t* comp = s.comp[s.sorted];
    hipLaunchKernelGGL(k_bf_emit, dim3((unsigned)nblocks), dim3(RS_BLOCK), 0, stream, comp, s.vsorted, s.n, s.kbits, s.ibits, s.pmin, s.kmin, mode, combine,
                       (const uint32_t*)s.cnt_c, (const uint32_t*)s.cnt_p, out_keys, out_vals, part_keys, mode == 2 ? s.scell : (uint32_t*)nullptr,
                       dctl, static_cast<LongRun*>(s.queue), der_comp, der_val, s.pbits);
    hipLaunchKernelGGL(k_fold_long, dim3(256), dim3(64), 0, stream, comp, s.vsorted, s.n, s.ibits, combine, (const BuildCtl*)dctl,
                       (const LongRun*)s.queue, out_vals, der_val);
    if (mode == 2 && nparts_explicit > 0)
        hipLaunchKernelGGL(k_emit_sems, dim3((unsigned)((nparts_explicit + 255) / 256)), dim3(256), 0, stream, comp, s.kbits, s.ibits, s.pmin,
                           (const uint32_t*)s.scell, s.n, nparts_explicit, out_keys, out_vals);
    hipError_t e = hipGetLastError();
    if (!wait_and_free) return e;            // the caller enqueues more behind the emit and calls build_abort(s) after its own stream wait
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    free_scratch(s);
    return e;
}

void build_abort(BuildScratch& s) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    free_scratch(s);
}

// ---- general path, host side ----
// one stable LSD sort of (comp, payload) over `bits` bits with the kernels of the composite path; returns the buffer index of the result
static hipError_t wide_sort(BuildScratch& s, int& cur, int bits, int64_t nblocks, hipStream_t stream) {
    const size_t lds_bytes = (size_t)RS_TILE * (sizeof(uint64_t) + sizeof(double));
    const dim3 grid((unsigned)nblocks), block(RS_BLOCK);
    const int npass = (bits + 7) / 8;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = 8 * pass;
        uint32_t* dtot = s.ghist + pass * RS_BINS;
        hipLaunchKernelGGL(k_rs_hist, grid, block, 0, stream, (const uint64_t*)s.comp[cur], s.n, shift, s.hist, nblocks);
        hipLaunchKernelGGL(k_rs_scan, dim3(RS_BINS), block, 0, stream, s.hist, dtot, nblocks);
        hipLaunchKernelGGL(k_rs_scatter<true>, grid, block, lds_bytes, stream, (const uint64_t*)s.comp[cur], (const double*)s.val[cur], s.n, shift,
                           (const uint32_t*)s.hist, (const uint32_t*)dtot, s.comp[1 - cur], s.val[1 - cur], nblocks);
        cur = 1 - cur;
    }
    return hipGetLastError();
}

static hipError_t prepare_wide(const int64_t* d_part, const int64_t* d_key, int64_t nnz, BuildScratch& s, int64_t counts[2], hipStream_t stream,
                               int64_t kmin, int kbits, int64_t pmin, int pbits) {
    const size_t n = (size_t)nnz;
    const int64_t nblocks = (nnz + RS_TILE - 1) / RS_TILE;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    BCHK(pinned_ctl_get(&s.h_ctl));
    {
        const size_t a4 = up(n * 4), a8 = up(n * 8), b_ctl = up(sizeof(BuildCtl)), b_gh = up(8 * RS_BINS * 4 + 64),
                     b_hist = up((size_t)RS_BINS * nblocks * 4), b_cnt = up((size_t)(nblocks + 1) * 4);
        BCHK(pool_alloc(&s.base, b_ctl + b_gh + 6 * a8 + 5 * a4 + b_hist + 2 * b_cnt));
        char* q = static_cast<char*>(s.base);
        auto take = [&q](size_t b) { char* r = q; q += b; return r; };
        s.d_ctl = take(b_ctl); s.ghist = (uint32_t*)take(b_gh);
        s.comp[0] = (uint64_t*)take(a8); s.comp[1] = (uint64_t*)take(a8); s.val[0] = (double*)take(a8); s.val[1] = (double*)take(a8);
        s.k2 = (int64_t*)take(a8); s.p2 = (int64_t*)take(a8);
        s.idx2 = (uint32_t*)take(a4); s.fpart = (uint32_t*)take(a4); s.fcell = (uint32_t*)take(a4); s.spart = (uint32_t*)take(a4); s.scell = (uint32_t*)take(a4);
        s.hist = (uint32_t*)take(b_hist); s.cnt_c = (uint32_t*)take(b_cnt); s.cnt_p = (uint32_t*)take(b_cnt);
    }
    BuildCtl* dctl = static_cast<BuildCtl*>(s.d_ctl);
    BuildCtl* hctl = static_cast<BuildCtl*>(s.h_ctl);
    static PerDeviceOnce once;
    const size_t lds_bytes = (size_t)RS_TILE * (sizeof(uint64_t) + sizeof(double));
    BCHK(once.run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(k_rs_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }));
    const unsigned gs = (unsigned)std::min<int64_t>((nnz + 255) / 256, 65535 * 4);
    int cur = 0;
    hipLaunchKernelGGL(k_wide_key0, dim3(gs), dim3(256), 0, stream, d_key, kmin, s.comp[0], s.val[0], nnz);
    BCHK(wide_sort(s, cur, kbits, nblocks, stream));                                                        // by key
    if (d_part != nullptr) {
        hipLaunchKernelGGL(k_wide_part, dim3(gs), dim3(256), 0, stream, d_part, pmin, (const double*)s.val[cur], s.comp[cur], nnz);
        BCHK(wide_sort(s, cur, std::max(pbits, 1), nblocks, stream));                                       // then by partition (stable)
    }
    hipLaunchKernelGGL(k_wide_finish, dim3(gs), dim3(256), 0, stream, d_part, d_key, (const double*)s.val[cur], s.idx2, s.k2, s.p2, nnz);
    const unsigned blocks = (unsigned)((nnz + 255) / 256);
    hipLaunchKernelGGL(k_flags, dim3(blocks), dim3(256), 0, stream, d_part != nullptr ? (const int64_t*)s.p2 : (const int64_t*)nullptr, s.k2, s.fpart, s.fcell, nnz);
    hipLaunchKernelGGL(k_flag_tile_sums, dim3((unsigned)nblocks), dim3(RS_BLOCK), 0, stream, (const uint32_t*)s.fpart, (const uint32_t*)s.fcell, nnz, s.cnt_c, s.cnt_p);
    hipLaunchKernelGGL(k_bf_scan, dim3(1), dim3(1024), 0, stream, s.cnt_c, s.cnt_p, nblocks, dctl);
    hipLaunchKernelGGL(k_flag_scan_apply, dim3((unsigned)nblocks), dim3(RS_BLOCK), 0, stream, (const uint32_t*)s.fpart, (const uint32_t*)s.fcell, nnz,
                       (const uint32_t*)s.cnt_c, (const uint32_t*)s.cnt_p, s.spart, s.scell);
    BCHK(hipMemcpyAsync(hctl, dctl, sizeof(BuildCtl), hipMemcpyDeviceToHost, stream));
    BCHK(hipStreamSynchronize(stream));
    counts[0] = (int64_t)hctl->ncells; counts[1] = (int64_t)hctl->nparts;
    return hipGetLastError();
}

static hipError_t emit_wide(const double* d_val, int32_t combine, BuildScratch& s, KeyArr out_keys, double* out_vals, int64_t* part_keys, int mode,
                            int64_t nparts_explicit, hipStream_t stream, bool wait_and_free) {
    const unsigned blocks = (unsigned)((s.n + 255) / 256);
    hipLaunchKernelGGL(k_emit_wide, dim3(blocks), dim3(256), 0, stream, mode == 1 ? (const int64_t*)nullptr : s.p2, s.k2, s.idx2, d_val,
                       s.fpart, s.fcell, s.spart, s.scell, s.n, combine, out_keys, out_vals, part_keys, mode);
    if (mode == 2 && nparts_explicit > 0)
        hipLaunchKernelGGL(k_emit_sems_wide, dim3((unsigned)((nparts_explicit + 255) / 256)), dim3(256), 0, stream, s.p2, s.scell, s.n,
                           nparts_explicit, out_keys, out_vals);
    hipError_t e = hipGetLastError();
    if (!wait_and_free) return e;            // (enqueue only, like the composite path: the caller waits and calls build_abort)
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    free_scratch(s);
    return e;
}

}  // namespace dsa
