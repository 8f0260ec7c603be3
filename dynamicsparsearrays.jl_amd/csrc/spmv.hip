// csrc/spmv.hip — K-spmv: PCSR sparse matrix x dense vector on gfx950.
//
// Reproduces _mul / _mul_dyn_mat_col_loop! (src/operations.jl:62-135) behind the `*` methods
// (src/operations.jl:14-60) for a dense x (every index stored).  Two forms over ONE orientation
// P of the matrix (a MappedPackedCSC: slot array + semaphores + partition keys):
//
//   gather  : y[part_key(p)] = sum_{slots of partition p} val * x[key]      (no atomics inside a row)
//             used on the TWIN orientation: mat*v walks rowmajor, transpose(mat)*v walks colmajor.
//             Per output row the terms are added left to right in ascending key order — exactly
//             the reference's accumulation order — except for rows that straddle a tile boundary
//             (two or more partial sums joined by fp64 atomics; within the 1e-12 tolerance).
//   scatter : y[key] += x[part_key(p)] * val with fp64 atomics — the literal loop nest of the
//             reference on its own orientation (mat*v walks colmajor).
//
// Bound: HBM.  Algorithmic bytes per launch = 16*capacity + 8*nx + 8*ny: the flat
// scan streams every slot (gaps included — they are part of the bit-identical layout) once.
//
// Kernel shape: one 256-thread workgroup per 2048-slot tile; wave w owns 8 occupancy words, lane <->
// slot (one ballot-shaped word per iteration, coalesced 8-byte key/value streams).  Products are
// staged in LDS; semaphore slots are compacted per wave with ballot + popcount; each semaphore's
// owner lane then sums its segment from LDS in slot order.  The partition active at the tile start
// is found by a short backward ballot scan (fallback: bisection of the semaphore table).
#include "dsa_dev.h"

namespace dsa {

constexpr int SP_TILE = 2048;
constexpr int SP_BLOCK = 256;
constexpr int SP_WAVES = SP_BLOCK / 64;
constexpr int SP_PER_WAVE = SP_TILE / SP_WAVES;     // 512 slots = 8 words
constexpr int SP_WORDS_PER_WAVE = SP_PER_WAVE / 64;
constexpr int SP_LONG = 48;                          // segments longer than this are summed by a whole wave
constexpr int SP_MAXLONG = 64;
constexpr int SP_BACK_WORDS = 32;                    // backward ballot scan limit before the table bisection

__device__ __forceinline__ double wave_reduce_add_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// partition id (1-based) whose semaphore is the last one located at a 0-based slot < b0, or 0.
// Executed by one full wave.
__device__ int64_t carry_in_partition(const int64_t* __restrict__ keys, const double* __restrict__ vals,
                                      const uint64_t* __restrict__ occ, const int64_t* __restrict__ sems,
                                      int64_t table_len, int64_t b0) {
    const int lane = lane_id();
    int64_t w = (b0 >> 6) - 1;
    for (int it = 0; it < SP_BACK_WORDS && w >= 0; ++it, --w) {
        const uint64_t word = occ[w];
        bool issem = false;
        if ((word >> lane) & 1ull) issem = (keys[(w << 6) + lane] == SEM_KEY);
        const uint64_t b = __ballot(issem);
        if (b) {
            const int hi = 63 - __clzll(b);
            return (int64_t)vals[(w << 6) + hi];
        }
    }
    if (w < 0) return 0;
    // bisection over the semaphore table (positions increase with the partition id; 0 = tombstone)
    int64_t lo = 0, hi = table_len - 1, best = -1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) >> 1;
        int64_t i = mid;
        while (i >= lo && sems[i] == 0) --i;
        if (i < lo) { lo = mid + 1; continue; }
        if (sems[i] <= b0) { best = i; lo = mid + 1; } else { hi = i - 1; }
    }
    return best + 1;
}

struct LongSeg { int start, end; int64_t row; int atomic; };

template <bool SCATTER>
__global__ __launch_bounds__(SP_BLOCK) void k_spmv(const int64_t* __restrict__ keys, const double* __restrict__ vals,
                                                   const uint64_t* __restrict__ occ, int64_t capacity,
                                                   const int64_t* __restrict__ sems,
                                                   const int64_t* __restrict__ part_keys, int64_t table_len,
                                                   const double* __restrict__ x, int64_t nx,
                                                   double* __restrict__ y, int64_t ny, int pattern) {
    __shared__ double sP[SP_TILE];
    __shared__ uint16_t sSemList[SP_WAVES][SP_PER_WAVE];
    __shared__ int sSemCnt[SP_WAVES];
    __shared__ int64_t sCarry;
    __shared__ LongSeg sLong[SP_MAXLONG];
    __shared__ int sNLong;
    __shared__ uint64_t sSemBits[SP_TILE / 64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 shares an XCD, speed only),
    // so XCD g is given the g-th contiguous eighth of the slot array: rows that are neighbours stay on one L2, and a
    // matrix with any band / block structure gathers x from a range that its XCD already holds.
    const int64_t ntiles = (capacity + SP_TILE - 1) / SP_TILE;
    int64_t tile = blockIdx.x;
    if (ntiles >= 64 && !(pattern & 4)) {                 // (pattern bit 2: dev knob DSA_DBG_SPMV=4 keeps the identity map)
                                                          // grid = 8 * ceil(ntiles / 8): (xcd, i) -> xcd * per + i is onto [0, ntiles)
        const int64_t per = (ntiles + 7) / 8;
        tile = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (tile >= ntiles) return;
    }
    const int64_t b0 = tile * SP_TILE;
    const int tile_end = (int)((capacity - b0) < SP_TILE ? (capacity - b0) : SP_TILE);
    if (tid == 0) sNLong = 0;

    // ---- phase 1: stream the tile, products (gather) or raw values (scatter) -> LDS -----------------
    int64_t k[SP_WORDS_PER_WAVE];
    double v[SP_WORDS_PER_WAVE];
    bool bit[SP_WORDS_PER_WAVE];
#pragma unroll
    for (int j = 0; j < SP_WORDS_PER_WAVE; ++j) {
        const int ls = wv * SP_PER_WAVE + j * 64 + lane;
        const int64_t s = b0 + ls;
        bit[j] = false; k[j] = -1; v[j] = 0.0;
        if (ls < tile_end) {
            // streamed once: non-temporal, so that the slot streams do not evict x from the XCD's L2
            const uint64_t word = __builtin_nontemporal_load(occ + (s >> 6));
            bit[j] = (word >> lane) & 1ull;
            if (bit[j]) { k[j] = __builtin_nontemporal_load(keys + s); v[j] = __builtin_nontemporal_load(vals + s); }
        }
    }
    int nsem = 0;
#pragma unroll
    for (int j = 0; j < SP_WORDS_PER_WAVE; ++j) {
        const int ls = wv * SP_PER_WAVE + j * 64 + lane;
        const bool issem = bit[j] && k[j] == SEM_KEY;
        double p = 0.0;
        if (issem) p = v[j];
        else if (bit[j]) {
            if (SCATTER) p = v[j];
            else if (k[j] >= 1 && k[j] <= nx) {
                const double xv = x[k[j] - 1];
                // pattern pass (touched rows of _mul, src/operations.jl:101): count the cells whose x entry is stored
                p = (pattern & 3) == 1 ? (xv != 0.0 ? 1.0 : 0.0) : v[j] * xv;
            }
        }
        sP[ls] = p;
        const uint64_t sb = __ballot(issem);
        if (issem) sSemList[wv][nsem + popc64(sb & mask_lt(lane))] = (uint16_t)ls;
        nsem += popc64(sb);
        if (SCATTER && lane == 0) sSemBits[wv * SP_WORDS_PER_WAVE + j] = sb;
    }
    if (lane == 0) sSemCnt[wv] = nsem;
    __syncthreads();

    int cnt[SP_WAVES];
    int total = 0;
#pragma unroll
    for (int w = 0; w < SP_WAVES; ++w) { cnt[w] = sSemCnt[w]; total += cnt[w]; }
    int first_sem = tile_end;
#pragma unroll
    for (int w = SP_WAVES - 1; w >= 0; --w) if (cnt[w] > 0) first_sem = sSemList[w][0];

    // the partition that owns the slots in front of the first semaphore of the tile
    if (wv == 0 && (first_sem > 0)) {
        const int64_t c = carry_in_partition(keys, vals, occ, sems, table_len, b0);
        if (lane == 0) sCarry = c;
    }

    if ((pattern & 3) == 2) return;      // dev ablation (DSA_DBG_SPMV=2): stream + gather only, no segmented sums
    if (!SCATTER) {
        // ---- phase 2 (gather): one lane per semaphore sums its segment in slot order -------------
        for (int j = tid; j < total; j += SP_BLOCK) {
            int w = 0, idx = j;
            while (idx >= cnt[w]) { idx -= cnt[w]; ++w; }
            const int a = sSemList[w][idx];
            int end = tile_end;
            if (idx + 1 < cnt[w]) end = sSemList[w][idx + 1];
            else {
                for (int w2 = w + 1; w2 < SP_WAVES; ++w2) if (cnt[w2] > 0) { end = sSemList[w2][0]; break; }
            }
            const int64_t id = (int64_t)sP[a];
            const int64_t row = part_keys[id - 1];
            const int is_last = (j == total - 1);
            if (end - a - 1 > SP_LONG) {
                const int e = atomicAdd(&sNLong, 1);
                sLong[e] = LongSeg{a + 1, end, row, is_last};
            } else if (row >= 1 && row <= ny) {
                // left-to-right sum of the segment (the reference's accumulation order); four LDS reads in flight
                double sum = 0.0;
                int s = a + 1;
                for (; s + 3 < end; s += 4) {
                    const double t0 = sP[s], t1 = sP[s + 1], t2 = sP[s + 2], t3 = sP[s + 3];
                    sum = sum + t0; sum = sum + t1; sum = sum + t2; sum = sum + t3;
                }
                for (; s < end; ++s) sum = sum + sP[s];
                if (is_last) atomicAdd(&y[row - 1], sum);    // the row may continue in the next tile
                else y[row - 1] = sum;
            }
        }
        __syncthreads();
        // head: slots in front of the first semaphore belong to the carried-in partition
        if (wv == 0 && first_sem > 0) {
            const int64_t c = sCarry;
            if (c > 0) {
                double sum = 0.0;
                for (int s = lane; s < first_sem; s += 64) sum += sP[s];
                sum = wave_reduce_add_f64(sum);
                const int64_t row = part_keys[c - 1];
                if (lane == 0 && row >= 1 && row <= ny) atomicAdd(&y[row - 1], sum);
            }
        }
        // long segments: a whole wave each
        const int nlong = sNLong;
        for (int e = wv; e < nlong; e += SP_WAVES) {
            const LongSeg L = sLong[e];
            double sum = 0.0;
            for (int s = L.start + lane; s < L.end; s += 64) sum += sP[s];
            sum = wave_reduce_add_f64(sum);
            if (lane == 0 && L.row >= 1 && L.row <= ny) {
                if (L.atomic) atomicAdd(&y[L.row - 1], sum);
                else y[L.row - 1] = sum;
            }
        }
    } else {
        // ---- phase 2 (scatter): every cell finds the semaphore that precedes it -------------------
        __shared__ int sLastBefore[SP_TILE / 64];
        __syncthreads();
        if (tid < SP_TILE / 64) {
            int last = -1;
            for (int w = 0; w < tid; ++w) {
                const uint64_t b = sSemBits[w];
                if (b) last = w * 64 + 63 - __clzll(b);
            }
            sLastBefore[tid] = last;
        }
        __syncthreads();
        const int64_t carry = (first_sem > 0) ? sCarry : 0;
#pragma unroll
        for (int j = 0; j < SP_WORDS_PER_WAVE; ++j) {
            const int wi = wv * SP_WORDS_PER_WAVE + j;
            if (bit[j] && k[j] != SEM_KEY) {
                const uint64_t mine = sSemBits[wi] & mask_lt(lane);
                const int owner = mine ? (wi * 64 + 63 - __clzll(mine)) : sLastBefore[wi];
                const int64_t id = owner >= 0 ? (int64_t)sP[owner] : carry;
                if (id > 0) {
                    const int64_t col = part_keys[id - 1];
                    if (col >= 1 && col <= nx && k[j] >= 1 && k[j] <= ny) atomicAdd(&y[k[j] - 1], x[col - 1] * v[j]);
                }
            }
        }
    }
}

// ---- sparse x, driven by x's stored entries ------------------------------------------------------------------------
// The literal shape of _mul (src/operations.jl:107-135): for every stored (j, x_j) locate the column partition of j
// (64-ary search of the sorted live column keys, one wave per entry), walk its slot range — semaphore+1 .. next live
// semaphore-1 (src/operations.jl:81-95) — and accumulate x_j * coeff into y[row] with fp64 atomics; every touched
// row is flagged (a row whose products cancel or are zero still belongs to the result, src/operations.jl:101).
// Work is proportional to the cells of the matched columns, not to the capacity.
__global__ __launch_bounds__(256) void k_spmv_xdriven(const int64_t* __restrict__ keys, const double* __restrict__ vals,
                                                      const uint64_t* __restrict__ occ, int64_t capacity,
                                                      const int64_t* __restrict__ sems, const int64_t* __restrict__ col_keys,
                                                      const uint8_t* __restrict__ col_live, int64_t table_len,
                                                      const int64_t* __restrict__ xi, const double* __restrict__ xv, int64_t nx,
                                                      double* __restrict__ y, uint8_t* __restrict__ touched, int64_t ny) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= nx) return;
    const int64_t col = xi[e];
    const double xval = xv[e];
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
    int64_t pos = b ? L + __ffsll((unsigned long long)b) - 1 : H;
    while (pos > 0 && !col_live[pos - 1]) --pos;
    if (pos == 0 || col_keys[pos - 1] != col) return;            // no such column: x entry skipped (src/operations.jl:76-79)
    const int64_t from = sems[pos - 1] + 1;
    int64_t nxt = pos + 1;
    while (nxt <= table_len && sems[nxt - 1] == 0) ++nxt;
    const int64_t to = nxt <= table_len ? sems[nxt - 1] - 1 : capacity;
    for (int64_t s = from + lane; s <= to; s += 64) {
        if ((occ[(s - 1) >> 6] >> ((s - 1) & 63)) & 1ull) {
            const int64_t row = keys[s - 1];
            if (row >= 1 && row <= ny) {
                atomicAdd(&y[row - 1], xval * vals[s - 1]);
                touched[row - 1] = 1;
            }
        }
    }
}

hipError_t launch_spmv_xdriven(const int64_t* keys, const double* vals, const uint64_t* occ, int64_t capacity,
                               const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live, int64_t table_len,
                               const int64_t* xi, const double* xv, int64_t nx, double* y, uint8_t* touched, int64_t ny,
                               hipStream_t stream) {
    hipError_t e = hipMemsetAsync(y, 0, (size_t)ny * sizeof(double), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(touched, 0, (size_t)ny, stream);
    if (e != hipSuccess) return e;
    if (nx > 0)
        hipLaunchKernelGGL(k_spmv_xdriven, dim3((unsigned)((nx + 3) / 4)), dim3(256), 0, stream, keys, vals, occ, capacity, sems,
                           col_keys, col_live, table_len, xi, xv, nx, y, touched, ny);
    return hipGetLastError();
}

#include <cstdlib>
static hipError_t launch_spmv(bool scatter, int pattern, const int64_t* keys, const double* vals, const uint64_t* occ, int64_t capacity,
                              const int64_t* sems, const int64_t* part_keys, int64_t table_len, const double* x, int64_t nx,
                              double* y, int64_t ny, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(y, 0, (size_t)ny * sizeof(double), stream);
    if (e != hipSuccess) return e;
    { static const char* dbg = getenv("DSA_DBG_SPMV"); if (dbg && pattern == 0) pattern = atoi(dbg); }
    const int64_t ntiles = (capacity + SP_TILE - 1) / SP_TILE;
    const int64_t grid = ntiles >= 64 ? 8 * ((ntiles + 7) / 8) : ntiles;     // see the XCD-aware mapping in k_spmv
    if (scatter)
        hipLaunchKernelGGL(k_spmv<true>, dim3((unsigned)grid), dim3(SP_BLOCK), 0, stream, keys, vals, occ, capacity, sems,
                           part_keys, table_len, x, nx, y, ny, 0);
    else
        hipLaunchKernelGGL(k_spmv<false>, dim3((unsigned)grid), dim3(SP_BLOCK), 0, stream, keys, vals, occ, capacity, sems,
                           part_keys, table_len, x, nx, y, ny, pattern);
    return hipGetLastError();
}

hipError_t launch_spmv_gather(const int64_t* keys, const double* vals, const uint64_t* occ, int64_t capacity,
                              const int64_t* sems, const int64_t* part_keys, const uint8_t*, int64_t table_len,
                              const double* x, int64_t nx, double* y, int64_t ny, int pattern, hipStream_t stream) {
    return launch_spmv(false, pattern, keys, vals, occ, capacity, sems, part_keys, table_len, x, nx, y, ny, stream);
}
hipError_t launch_spmv_scatter(const int64_t* keys, const double* vals, const uint64_t* occ, int64_t capacity,
                               const int64_t* sems, const int64_t* part_keys, const uint8_t*, int64_t table_len,
                               const double* x, int64_t nx, double* y, int64_t ny, hipStream_t stream) {
    return launch_spmv(true, 0, keys, vals, occ, capacity, sems, part_keys, table_len, x, nx, y, ny, stream);
}

}  // namespace dsa
