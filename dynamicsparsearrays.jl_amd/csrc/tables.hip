// csrc/tables.hip — grid-wide merge of the pending partition-table entries of a MappedPackedCSC.
//
// addpartition!(pcsc, prev) in the middle of the tables (src/pcsr.jl:114-146) shifts semaphores[] / col_keys[] one entry to the
// right and rewrites the id stored in every later semaphore cell.  The write paths (sequencer.hip, parbatch.hip) defer that:
// a new partition gets the next free id at the END of the tables (Ctl::n_pending entries, arrival order) and its semaphore cell
// goes where the reference puts it; ids are only labels while a batch is running.  This file brings the tables back to key
// order — the reference's numbering — in ONE pass over the table with the whole chip instead of the sequencer's single
// workgroup (a 100k-row rowmajor twin: 137 us per merge there, a few us here; 10M rows: milliseconds there):
//
//   k_merge_sort    one workgroup: ranks the K <= 1024 pending keys (counting in LDS), destination of pending rank r =
//                   (#sorted keys below it) + r, the smallest destination i_min bounds the part of the table that moves
//   k_merge_move    grid: sorted entry i >= i_min moves up by the number of pending keys below its key (binary search in the
//                   sorted pending keys held in LDS) into the scratch tables, pending entries go to their destinations, and
//                   every semaphore cell whose id changed is rewritten once (vals[semaphore slot] = new id)
//   k_merge_commit  grid: scratch -> tables for [i_min, table_len), Ctl::n_pending = 0
//
// All three are stream-ordered launches without a host wait; with nothing pending they return at once.
#include "dsa_dev.h"

namespace dsa {

constexpr int MG_PEND_MAX = TABLE_PEND_MAX;
static_assert(MG_PEND_MAX <= 1024, "k_merge_sort ranks the pending keys with one thread each");

__global__ __launch_bounds__(1024) void k_merge_sort(const int64_t* sems, const int64_t* col_keys, const Ctl* ctl, TableMerge tm) {
    __shared__ int64_t sKey[MG_PEND_MAX];
    const int64_t K = ctl->n_pending, table_len = ctl->table_len, ns = table_len - K;
    const int r = threadIdx.x;
    if (K <= 0 || K > MG_PEND_MAX) {
        if (r == 0) { tm.hdr[0] = 0; tm.hdr[1] = 0; tm.hdr[2] = K > MG_PEND_MAX ? 1 : 0; }
        return;
    }
    int64_t key = 0;
    if (r < K) { key = col_keys[ns + r]; sKey[r] = key; }
    __syncthreads();
    if (r >= K) return;
    int rank = 0;
    for (int j = 0; j < (int)K; ++j) rank += sKey[j] < key ? 1 : 0;          // keys are distinct
    int64_t lo = 0, hi = ns;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (col_keys[mid] < key) lo = mid + 1; else hi = mid; }
    tm.pkey[rank] = key;
    tm.pdst[rank] = lo + rank;
    tm.psem[rank] = sems[ns + r];
    if (rank == 0) { tm.hdr[0] = K; tm.hdr[1] = lo; tm.hdr[2] = 0; }
}

__global__ __launch_bounds__(256) void k_merge_move(const int64_t* sems, const int64_t* col_keys, double* vals, const Ctl* ctl, TableMerge tm) {
    __shared__ int64_t sKey[MG_PEND_MAX];
    const int K = (int)tm.hdr[0];
    if (K == 0) return;
    const int64_t i_min = tm.hdr[1], ns = ctl->table_len - K;
    for (int j = threadIdx.x; j < K; j += blockDim.x) sKey[j] = tm.pkey[j];
    __syncthreads();
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i_min + gtid; i < ns; i += stride) {
        const int64_t ck = col_keys[i], sp = sems[i];
        int lo = 0, hi = K;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sKey[mid] < ck) lo = mid + 1; else hi = mid; }
        const int64_t d = i + lo;                                               // lo >= 1: every entry from i_min on has a pending key below it
        tm.sems2[d] = sp; tm.keys2[d] = ck;
        if (sp != 0) vals[sp - 1] = (double)(d + 1);
    }
    for (int64_t r = gtid; r < K; r += stride) {
        const int64_t d = tm.pdst[r], sp = tm.psem[r];
        tm.sems2[d] = sp; tm.keys2[d] = sKey[r];
        vals[sp - 1] = (double)(d + 1);
    }
}

__global__ __launch_bounds__(256) void k_merge_commit(int64_t* sems, int64_t* col_keys, uint8_t* col_live, Ctl* ctl, TableMerge tm) {
    const int K = (int)tm.hdr[0];
    if (K == 0) return;
    const int64_t i_min = tm.hdr[1], table_len = ctl->table_len;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t d = i_min + gtid; d < table_len; d += stride) { sems[d] = tm.sems2[d]; col_keys[d] = tm.keys2[d]; col_live[d] = 1; }
    if (gtid == 0) ctl->n_pending = 0;
}

// tm.sems2 / tm.keys2 hold table_cap entries, tm.pkey / pdst / psem MG_PEND_MAX, tm.hdr 4
hipError_t launch_table_merge(int64_t* sems, int64_t* col_keys, uint8_t* col_live, double* vals, Ctl* ctl, TableMerge tm, int64_t table_cap,
                              hipStream_t stream) {
    int64_t blocks = (table_cap + 1023) / 1024;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_merge_sort, dim3(1), dim3(1024), 0, stream, sems, col_keys, ctl, tm);
    hipLaunchKernelGGL(k_merge_move, dim3((unsigned)blocks), dim3(256), 0, stream, sems, col_keys, vals, ctl, tm);
    hipLaunchKernelGGL(k_merge_commit, dim3((unsigned)blocks), dim3(256), 0, stream, sems, col_keys, col_live, ctl, tm);
    return hipGetLastError();
}

}  // namespace dsa
