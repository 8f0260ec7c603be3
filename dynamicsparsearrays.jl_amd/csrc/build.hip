// csrc/build.hip — K-build: the bulk (fill-mode flush) constructor on the device.
//
// Reproduces _dynamicsparse (src/pcsr.jl:354-431) + the PackedCSC constructor's cell stream
// (src/pcsr.jl:26-63) for one orientation: sort the (partition, key) pairs, combine duplicates, count the
// cells per partition and emit the ordered stream  [sem(0, id), entries...]  per partition, plus the
// partition keys.  The stream is written straight into the PMA's slot buffer; the full-array spread
// (src/pma.jl:42-55) is then one k_move<true> launch (rebalance.hip).
//
//   sort      : two stable LSD radix sorts (rocPRIM device_radix_sort, 64-bit signed keys) — by key, then by
//               partition — carrying the input index, so equal (partition, key) pairs stay in INPUT order.
//               The reference sorts with an unstable QuickSort (src/pcsr.jl:360); input order is one of its
//               legal outcomes and makes the Float64 fold of duplicates deterministic.
//   flags     : new-partition / new-cell flags from neighbour compares (coalesced).
//   scans     : two inclusive scans (rocPRIM device_scan) give the partition id and the cell rank.
//   emit      : every first-of-run lane folds its duplicate run left to right (src/pcsr.jl:374-375) and
//               writes its cell at  rank-1 + partition_id  ; every first-of-partition lane writes the
//               semaphore cell (0, id) at  rank-1 + id-1  and the partition key.
// Bound: HBM (sort passes dominate: 8 B key + 4 B index, 2 x 8 digit passes).
#include "dsa_dev.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace dsa {

__global__ void k_iota(uint32_t* idx, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}
__global__ void k_gather_i64(const int64_t* __restrict__ src, const uint32_t* __restrict__ idx, int64_t* __restrict__ dst, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_flags(const int64_t* __restrict__ part, const int64_t* __restrict__ key, uint32_t* __restrict__ fpart,
                        uint32_t* __restrict__ fcell, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    // part == nullptr: a plain vector (one implicit partition, no semaphore cells)
    const bool np = part == nullptr ? (i == 0) : ((i == 0) || part[i] != part[i - 1]);
    fpart[i] = np ? 1u : 0u;
    fcell[i] = (np || key[i] != key[i - 1]) ? 1u : 0u;
}
__global__ void k_emit(const int64_t* __restrict__ part, const int64_t* __restrict__ key, const uint32_t* __restrict__ idx,
                       const double* __restrict__ val, const uint32_t* __restrict__ fpart, const uint32_t* __restrict__ fcell,
                       const uint32_t* __restrict__ spart, const uint32_t* __restrict__ scell, int64_t n, int32_t combine,
                       KeyArr out_keys, double* __restrict__ out_vals, int64_t* __restrict__ part_keys,
                       int mode) {
    // mode 0: mapped partitions (ids = rank of the distinct partition keys, semaphore emitted by the first cell)
    // mode 1: plain vector (no semaphores)   mode 2: explicit partition ids 1..P in `part` (semaphores by k_emit_sems)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!fcell[i]) return;
    const int64_t pid = mode == 0 ? (int64_t)spart[i] : (mode == 1 ? 0 : part[i]);   // semaphore cells in front of this cell
    const int64_t rank = scell[i];           // 1-based rank among the distinct cells
    double acc = val[idx[i]];
    for (int64_t j = i + 1; j < n && !fcell[j]; ++j) {     // left fold of the duplicates, input order
        const double v = val[idx[j]];
        acc = combine == 0 ? acc + v : (combine == 1 ? acc * v : v);
    }
    const int64_t pos = rank - 1 + pid;
    out_keys[pos] = key[i];
    out_vals[pos] = acc;
    if (mode == 0 && fpart[i]) {
        out_keys[pos - 1] = SEM_KEY;
        out_vals[pos - 1] = (double)pid;
        part_keys[pid - 1] = part[i];
    }
}

// mode 2: semaphore cell of every partition p = 1..P (empty partitions included, src/pcsr.jl:36-41):
// position = (#distinct cells of partitions < p) + p - 1
__global__ void k_emit_sems(const int64_t* __restrict__ part_sorted, const uint32_t* __restrict__ scell, int64_t n, int64_t nparts,
                            KeyArr out_keys, double* __restrict__ out_vals) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x + 1;
    if (p > nparts) return;
    int64_t lo = 0, hi = n;                     // first index with part >= p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (part_sorted[mid] < p) lo = mid + 1; else hi = mid;
    }
    const int64_t cells_before = lo == 0 ? 0 : (int64_t)scell[lo - 1];
    const int64_t pos = cells_before + p - 1;
    out_keys[pos] = SEM_KEY;
    out_vals[pos] = (double)p;
}

// The scratch of a build is ONE allocation carved into its twelve arrays: twelve hipMalloc + hipFree pairs cost ~2 ms of the 5.3 ms a
// 10 M-triple orientation took.  (Stream-ordered allocation from the default pool — hipMallocAsync with a raised release
// threshold — was tried for it: 4 ms builds most of the time, but stalls of 120-150 ms in hipMallocAsync or in the next plain
// hipMalloc when the two orientations build side by side on two host threads.  Dropped.)
static void free_scratch(BuildScratch& s) {
    if (s.base) (void)hipFree(s.base);
    s = BuildScratch();
}

#define BCHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { free_scratch(s); return _e; } } while (0)

// Phase 1: sort + flags + scans.  d_part / d_key / d_val: the nnz input triples in HBM.
// Returns the number of distinct cells and of partitions through counts[0..1] (host).  The scratch stays alive
// for phase 2 (build_emit), which writes n_cells + n_parts stream cells.
hipError_t build_prepare(const int64_t* d_part, const int64_t* d_key, int64_t nnz, BuildScratch& s, int64_t counts[2],
                         hipStream_t stream) {
    s.n = nnz; s.stream = stream;
    const size_t n = (size_t)nnz;
    size_t t1 = 0, t2 = 0;
    BCHK(rocprim::radix_sort_pairs(nullptr, t1, d_key, s.k1, s.idx0, s.idx1, n, 0, 64, stream));
    BCHK(rocprim::inclusive_scan(nullptr, t2, s.fpart, s.spart, n, rocprim::plus<uint32_t>(), stream));
    s.temp_bytes = t1 > t2 ? t1 : t2;
    {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t a4 = up(n * 4), a8 = up(n * 8);
        BCHK(hipMalloc(&s.base, 7 * a4 + 4 * a8 + up(s.temp_bytes)));
        char* q = static_cast<char*>(s.base);
        auto take = [&q](size_t b) { char* r = q; q += b; return r; };
        s.idx0 = (uint32_t*)take(a4); s.idx1 = (uint32_t*)take(a4); s.idx2 = (uint32_t*)take(a4);
        s.fpart = (uint32_t*)take(a4); s.fcell = (uint32_t*)take(a4); s.spart = (uint32_t*)take(a4); s.scell = (uint32_t*)take(a4);
        s.k1 = (int64_t*)take(a8); s.p1 = (int64_t*)take(a8); s.p2 = (int64_t*)take(a8); s.k2 = (int64_t*)take(a8);
        s.temp = take(up(s.temp_bytes));
    }
    const unsigned blocks = (unsigned)((nnz + 255) / 256);
    hipLaunchKernelGGL(k_iota, dim3(blocks), dim3(256), 0, stream, s.idx0, nnz);
    size_t tb = s.temp_bytes;
    BCHK(rocprim::radix_sort_pairs(s.temp, tb, d_key, s.k1, s.idx0, s.idx1, n, 0, 64, stream));           // by key
    if (d_part != nullptr) {
        hipLaunchKernelGGL(k_gather_i64, dim3(blocks), dim3(256), 0, stream, d_part, s.idx1, s.p1, nnz);
        tb = s.temp_bytes;
        BCHK(rocprim::radix_sort_pairs(s.temp, tb, s.p1, s.p2, s.idx1, s.idx2, n, 0, 64, stream));       // then by partition (stable)
        hipLaunchKernelGGL(k_gather_i64, dim3(blocks), dim3(256), 0, stream, d_key, s.idx2, s.k2, nnz);
        hipLaunchKernelGGL(k_flags, dim3(blocks), dim3(256), 0, stream, s.p2, s.k2, s.fpart, s.fcell, nnz);
    } else {                                                                                               // vector: keys only
        BCHK(hipMemcpyAsync(s.idx2, s.idx1, n * 4, hipMemcpyDeviceToDevice, stream));
        BCHK(hipMemcpyAsync(s.k2, s.k1, n * 8, hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(k_flags, dim3(blocks), dim3(256), 0, stream, (const int64_t*)nullptr, s.k2, s.fpart, s.fcell, nnz);
    }
    tb = s.temp_bytes;
    BCHK(rocprim::inclusive_scan(s.temp, tb, s.fpart, s.spart, n, rocprim::plus<uint32_t>(), stream));
    tb = s.temp_bytes;
    BCHK(rocprim::inclusive_scan(s.temp, tb, s.fcell, s.scell, n, rocprim::plus<uint32_t>(), stream));
    uint32_t last[2] = {0, 0};
    BCHK(hipMemcpyAsync(&last[0], s.scell + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    BCHK(hipMemcpyAsync(&last[1], s.spart + (n - 1), 4, hipMemcpyDeviceToHost, stream));
    BCHK(hipStreamSynchronize(stream));
    counts[0] = last[0]; counts[1] = last[1];
    return hipGetLastError();
}

hipError_t build_emit(const double* d_val, int32_t combine, BuildScratch& s, KeyArr out_keys, double* out_vals,
                      int64_t* part_keys, int mode, int64_t nparts_explicit, hipStream_t stream) {
    const unsigned blocks = (unsigned)((s.n + 255) / 256);
    hipLaunchKernelGGL(k_emit, dim3(blocks), dim3(256), 0, stream, mode == 1 ? (const int64_t*)nullptr : s.p2, s.k2, s.idx2, d_val,
                       s.fpart, s.fcell, s.spart, s.scell, s.n, combine, out_keys, out_vals, part_keys, mode);
    if (mode == 2 && nparts_explicit > 0)
        hipLaunchKernelGGL(k_emit_sems, dim3((unsigned)((nparts_explicit + 255) / 256)), dim3(256), 0, stream, s.p2, s.scell, s.n,
                           nparts_explicit, out_keys, out_vals);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    free_scratch(s);
    return e;
}

void build_abort(BuildScratch& s) { free_scratch(s); }

}  // namespace dsa
