/* include/dsa.h — C ABI of libdsa_hip.so, the MI355X (gfx950) packed-memory-array /
 * packed-CSR engine.  This is the drop-in boundary a Julia host module `ccall`s
 * (see INTEGRATION.md for the binding).  The reference package
 * (atoptima/DynamicSparseArrays.jl v0.7.2) has no FFI of its own: each entry point
 * below replaces the Julia method cited next to it (paths relative to the
 * reference checkout).
 *
 * Conventions
 *   - K = L = Int64, T = Float64; semaphore key is 0
 *     (src/pcsr.jl:23), so 0 is rejected as a row/column key of a matrix.
 *   - All indices / positions crossing the ABI are 1-BASED, like the reference.
 *   - Inputs are borrowed host pointers valid for the call only; outputs are
 *     caller-allocated host buffers with explicit capacities.  `*_dev` entry points
 *     take DEVICE pointers (HBM-resident) and enqueue on the handle's stream.
 *   - Every function returns an int32 status; no exceptions cross the boundary.
 *     dsa_last_error_message() returns the text of the last failure on this thread.
 *   - A handle is single-writer / not thread-safe (as the reference).  All slot
 *     storage, occupancy bitmaps, semaphore and column-key tables live in HBM.
 *   - There is NO CPU fallback: every data-path operation runs HIP kernels and
 *     fails with DSA_EHIP if no gfx950 device is usable.
 */
#ifndef DSA_H
#define DSA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes; each mirrors a reference exception site */
enum {
    DSA_OK = 0,
    DSA_EARG = 1,      /* ArgumentError: src/vector.jl:49-50, src/pcsr.jl:208,439-442, src/views.jl:12 */
    DSA_EBOUNDS = 2,   /* BoundsError:   src/pcsr.jl:190, src/moves.jl:10-11 */
    DSA_EDELETED = 3,  /* "The partition has been deleted."  src/pcsr.jl:299 */
    DSA_EFULL = 4,     /* "No empty cell to insert a new element."  src/writes.jl:39 */
    DSA_EMODE = 5,     /* fill-mode misuse: src/matrix.jl:73,84,96,105,127 ; src/buffer.jl:13 */
    DSA_EASSERT = 6,   /* reference @assert sites: src/pcsr.jl:124,132,173,182 */
    DSA_EHIP = 7,      /* HIP runtime failure / no device */
    DSA_ECAP = 8,      /* caller-provided output buffer too small */
    DSA_EKEY = 9,      /* reserved key 0 used as a matrix row/column key */
    DSA_ERCCL = 10     /* RCCL failure / librccl.so not loadable (dsa_comm_*) */
};

enum { DSA_COMBINE_ADD = 0, DSA_COMBINE_MUL = 1, DSA_COMBINE_LAST = 2 };
enum { DSA_COLMAJOR = 0, DSA_ROWMAJOR = 1 };

/* index of the scalars returned by the *_info calls */
enum {
    DSA_INFO_CAPACITY = 0,        /* pma.capacity            src/pma.jl:9  */
    DSA_INFO_SEGMENT_CAPACITY = 1,/* pma.segment_capacity    src/pma.jl:10 */
    DSA_INFO_NB_SEGMENTS = 2,     /* pma.nb_segments         src/pma.jl:11 */
    DSA_INFO_NB_ELEMENTS = 3,     /* pma.nb_elements         src/pma.jl:12 */
    DSA_INFO_HEIGHT = 4,          /* pma.height              src/pma.jl:15 */
    DSA_INFO_NB_PARTITIONS = 5,   /* pcsc.nb_partitions      src/pcsr.jl:5  (vector: length n, src/vector.jl:2) */
    DSA_INFO_TABLE_LEN = 6,       /* length(pcsc.semaphores) == length(col_keys) (tombstones included) */
    DSA_INFO_STAT_WINDOW_SLOTS = 7,/* instrumentation: slots inside pack/spread windows so far */
    DSA_INFO_STAT_REBALANCES = 8, /* instrumentation: number of pack/spread windows so far */
    DSA_INFO_STAT_EXTENDS = 9,
    DSA_INFO_STAT_SHRINKS = 10,
    DSA_INFO_STAT_PAR_ROUNDS = 11, /* batch-parallel rounds / ops applied in parallel / ops applied by the sequencer */
    DSA_INFO_STAT_PAR_OPS = 12,
    DSA_INFO_STAT_SEQ_OPS = 13,
    DSA_INFO_STAT_SPMV_NOMEMSET = 14, /* instrumentation: gather SpMV launches over this orientation that needed no memset of y */
    DSA_INFO_HBM_BYTES = 15,          /* bytes of HBM the structure holds (slot buffers x 2, bitmaps, tables, merge scratch) */
    DSA_INFO_STAT_GRID_REBALANCES = 16, /* instrumentation: launches of the grid-wide pack/spread kernel (windows above 8192 slots, root, _extend!, _shrink!).
                                         * Windows an append run rebalances are replayed on the bitmap (csrc/appendmodel.hip) and moved by ONE K-permute
                                         * at the end of the run, whatever their size: they count in STAT_REBALANCES / STAT_WINDOW_SLOTS, not here. */
    DSA_INFO_COUNT = 17
};

typedef struct dsa_vec dsa_vec_t;    /* DynamicSparseVector   src/vector.jl:1-4   */
typedef struct dsa_pcsc dsa_pcsc_t;  /* PackedCSC             src/pcsr.jl:4-9     */
typedef struct dsa_mat dsa_mat_t;    /* DynamicSparseMatrix   src/matrix.jl:1-8   */

const char* dsa_last_error_message(void);
/* number of usable gfx950 devices and selection of the device new handles live on */
int32_t dsa_device_count(int32_t* count);
int32_t dsa_set_device(int32_t device);

/* ---------------- DynamicSparseVector (one PMA) ---------------- */
/* dynamicsparsevec(I, V, combine, n)  src/vector.jl:44-62 ; len < 0 => _guess_length(I) (:6) */
int32_t dsa_vec_create(const int64_t* keys, const double* vals, int64_t n, int32_t combine_op,
                       int64_t len, dsa_vec_t** out);
/* dynamicsparsevec(Int[], Float64[]) -> PackedMemoryArray(K,T)  src/pma.jl:86-91 */
int32_t dsa_vec_create_empty(dsa_vec_t** out);
int32_t dsa_vec_destroy(dsa_vec_t* h);
/* getindex(v, key)  src/vector.jl:73 -> src/pma.jl:189-193 */
int32_t dsa_vec_get(dsa_vec_t* h, int64_t key, double* out);
int32_t dsa_vec_get_batch(dsa_vec_t* h, const int64_t* keys, int64_t n, double* out);
/* setindex!(v, value, key)  src/vector.jl:76-81 -> src/pma.jl:196-213.  WRITE-COMBINED: the call queues the write on the
 * host; queued writes are applied in order by one device sequencer launch at the latest before the next call that observes
 * the vector (get, nnz, info, iteration, export, ...) or when 65536 are pending.  length(v) is updated immediately. */
int32_t dsa_vec_set(dsa_vec_t* h, int64_t key, double val);
/* n sequential setindex! calls, applied in order (sequential-equivalent batch) */
int32_t dsa_vec_set_batch(dsa_vec_t* h, const int64_t* keys, const double* vals, int64_t n);
/* nnz(v) src/vector.jl:88 ; length(v) :69 ; shrink_size!(v) :64 */
int32_t dsa_vec_nnz(dsa_vec_t* h, int64_t* out);
int32_t dsa_vec_len(dsa_vec_t* h, int64_t* out);
int32_t dsa_vec_shrink_size(dsa_vec_t* h);
/* iterate(v) src/vector.jl:71 / nonzeroinds+nonzeros :93-109 : stored entries in slot order */
int32_t dsa_vec_nonzeros(dsa_vec_t* h, int64_t* keys, double* vals, int64_t cap, int64_t* n_out);
/* v1 == v2  src/vector.jl:85-87 -> src/pma.jl:262-266 -> _arrays_equal src/pma.jl:236-260: equal length(v), equal nb_elements and
 * pairwise equal stored (key, value) tuples in slot order (layouts may differ).  *out = 1 / 0.  Both vectors are packed and
 * compared on the device; only the verdict crosses PCIe. */
int32_t dsa_vec_equal(dsa_vec_t* a, dsa_vec_t* b, int32_t* out);
/* alpha * a + beta * b as ascending (key, value) pairs: the SparseVector the reference's  v1 + v2  (1, 1),  v1 - v2  (1, -1) and
 * -v  (-1, 0, b = a) evaluate to through the AbstractSparseVector fallbacks over nonzeroinds / nonzeros
 * (src/vector.jl:93-109; test/functional/math.jl:53-94).  A key stored in both operands keeps one entry, dropped when its value
 * is zero; DSA_ECAP when cap < number of result entries (nnz(a) + nnz(b) always suffices).  Merge + compaction on the device. */
int32_t dsa_vec_axpby(dsa_vec_t* a, double alpha, dsa_vec_t* b, double beta, int64_t* keys, double* vals, int64_t cap, int64_t* n_out);
int32_t dsa_vec_info(dsa_vec_t* h, int64_t info[DSA_INFO_COUNT]);
/* parity probe / snapshot: slot array (keys, vals, occ[i] in {0,1}), cap >= capacity */
int32_t dsa_vec_export_layout(dsa_vec_t* h, int64_t* keys, double* vals, uint8_t* occ, int64_t cap);
/* _even_rebalance!(pma, 1, capacity, nb_elements) on the whole array  src/pma.jl:94-103
 * (benchmark / test hook for the full-window pack+spread kernel; layout-idempotent) */
int32_t dsa_vec_rebalance_root(dsa_vec_t* h);
/* Benchmark / test hook (no reference counterpart): puts the cells of the vector into a layout the reference reaches only
 * in the middle of an operation, WITHOUT changing which cells are stored or their order, so that the next
 * dsa_vec_rebalance_root can be timed on it:
 *   1 = pack!(array, 1, capacity, n): all cells in slots 1..n            (src/moves.jl:94-110; the source of _shrink!)
 *   2 = all cells in the LAST n slots (all gaps at the left: the window a run of appends leaves behind)
 *   3 = _extend!  (src/pma.jl:143-151: capacity x 2, root spread)        4 = _shrink! (src/pma.jl:153-161: capacity / 2)
 * 3 and 4 ignore the density thresholds; 1 and 2 leave a layout that violates them until the next root rebalance. */
int32_t dsa_vec_dev_relayout(dsa_vec_t* h, int32_t mode);

/* ---------------- PackedCSC (integer-indexed partitions) ---------------- */
/* PackedCSC(row_keys::Vector{Vector}, values::Vector{Vector}, combine)  src/pcsr.jl:26-63
 * given CSC-style: partition p holds entries [colptr[p], colptr[p+1]) (0-based offsets, nparts+1 entries) */
int32_t dsa_pcsc_create(const int64_t* colptr, int64_t nparts, const int64_t* row_keys,
                        const double* vals, int32_t combine_op, dsa_pcsc_t** out);
int32_t dsa_pcsc_create_empty(dsa_pcsc_t** out);                       /* src/pcsr.jl:65-68 */
int32_t dsa_pcsc_destroy(dsa_pcsc_t* h);
int32_t dsa_pcsc_get(dsa_pcsc_t* h, int64_t key, int64_t partition, double* out);   /* src/pcsr.jl:228-232 */
int32_t dsa_pcsc_set(dsa_pcsc_t* h, double val, int64_t key, int64_t partition);    /* src/pcsr.jl:294-310 */
int32_t dsa_pcsc_deletepartition(dsa_pcsc_t* h, int64_t partition);                 /* src/pcsr.jl:188-204 */
int32_t dsa_pcsc_nnz(dsa_pcsc_t* h, int64_t* out);                                  /* src/pcsr.jl:11 */
int32_t dsa_pcsc_nbpartitions(dsa_pcsc_t* h, int64_t* out);                         /* src/pcsr.jl:21 */
int32_t dsa_pcsc_info(dsa_pcsc_t* h, int64_t info[DSA_INFO_COUNT]);
/* semaphores[id] = slot of partition id's semaphore, 0 = nothing (tombstone) */
int32_t dsa_pcsc_export_layout(dsa_pcsc_t* h, int64_t* keys, double* vals, uint8_t* occ, int64_t cap,
                               int64_t* semaphores, int64_t table_cap);

/* ---------------- DynamicSparseMatrix (colmajor + rowmajor MappedPackedCSC, or fill buffer) ---------------- */
/* dynamicsparse(I, J, V, m, n)  src/matrix.jl:15-19 -> dynamicsparsecolmajor src/pcsr.jl:433-445 (x2) ;
 * m, n < 0 => _guess_length */
int32_t dsa_mat_create_from_coo(const int64_t* I, const int64_t* J, const double* V, int64_t nnz,
                                int64_t m, int64_t n, dsa_mat_t** out);
/* dynamicsparse(K, L, T; fill_mode)  src/matrix.jl:31-41 */
int32_t dsa_mat_create_empty(int32_t fill_mode, dsa_mat_t** out);
int32_t dsa_mat_destroy(dsa_mat_t* h);
/* setindex!(m, val, row, col)  src/matrix.jl:43-62 (fill mode: addelem! src/buffer.jl:20-31).  WRITE-COMBINED like
 * dsa_vec_set (size(m) is updated immediately); when the matrix holds deleted columns/rows — the only state in which a write
 * can fail in the reference — the write is applied before the call returns so the error surfaces here. */
int32_t dsa_mat_set(dsa_mat_t* h, double val, int64_t row, int64_t col);
/* n sequential setindex! calls in order.  On success the state equals that of the n calls.  When write k fails (only possible while
 * the matrix holds deleted columns / rows) the status, size(m) and the contents are those of the reference at its
 * exception (src/matrix.jl:43-62 updates colmajor, then rowmajor): both orientations hold writes [0, k), colmajor also holds write k when
 * it was the rowmajor statement that threw.  (The orientation that refused the write keeps its partition tables as they were; the
 * reference leaves them half shifted — a state nothing can continue from.) */
int32_t dsa_mat_set_batch(dsa_mat_t* h, const int64_t* I, const int64_t* J, const double* V, int64_t n);
/* getindex(m, row, col)  src/matrix.jl:64-68 */
int32_t dsa_mat_get(dsa_mat_t* h, int64_t row, int64_t col, double* out);
int32_t dsa_mat_get_batch(dsa_mat_t* h, const int64_t* I, const int64_t* J, int64_t n, double* out);
/* addrow!(matrix, row, colids, vals)  src/matrix.jl:113-124 (fill mode: src/buffer.jl:10-18) */
int32_t dsa_mat_addrow(dsa_mat_t* h, int64_t row, const int64_t* colids, const double* vals, int64_t n);
/* closefillmode!  src/matrix.jl:126-134 */
int32_t dsa_mat_closefillmode(dsa_mat_t* h);
/* deletecolumn! / deleterow!  src/matrix.jl:95-111 */
int32_t dsa_mat_deletecolumn(dsa_mat_t* h, int64_t col);
int32_t dsa_mat_deleterow(dsa_mat_t* h, int64_t row);
/* @view m[:, col] / @view m[row, :] iteration  src/matrix.jl:70-93, src/views.jl:15-35 */
int32_t dsa_mat_col_view(dsa_mat_t* h, int64_t col, int64_t* rows, double* vals, int64_t cap, int64_t* n_out);
int32_t dsa_mat_row_view(dsa_mat_t* h, int64_t row, int64_t* cols, double* vals, int64_t cap, int64_t* n_out);
/* the same views delivered into HBM: d_rows / d_vals are DEVICE arrays of cap entries; the cells are packed on the device and copied
 * device-to-device on the orientation's stream (dsa_mat_set_stream / dsa_mat_sync); *n_out is valid on return.  No cell crosses PCIe. */
int32_t dsa_mat_col_view_dev(dsa_mat_t* h, int64_t col, int64_t* d_rows, double* d_vals, int64_t cap, int64_t* n_out);
int32_t dsa_mat_row_view_dev(dsa_mat_t* h, int64_t row, int64_t* d_cols, double* d_vals, int64_t cap, int64_t* n_out);
/* m[:, col] / m[row, :] as a NEW dynamic sparse vector  (getindex(mpcsc, :, col) src/pcsr.jl:285-291 -> :247-259 ;
 * getindex(mpcsc, row, :) src/pcsr.jl:269-283): the stored entries of the column / row, length = largest key.  Device to device: the
 * view kernel packs the partition into the orientation's idle slot buffer and ONE spread launch writes the new vector's slot array
 * from there (PackedMemoryArray(elements), src/pma.jl:69-84); only the entry count and the largest key reach the host. */
int32_t dsa_mat_col_slice(dsa_mat_t* h, int64_t col, dsa_vec_t** out);
int32_t dsa_mat_row_slice(dsa_mat_t* h, int64_t row, dsa_vec_t** out);
/* nnz(m) src/matrix.jl:91 ; size(m) :92 ; nbpartitions(orientation) src/pcsr.jl:21-22 */
int32_t dsa_mat_nnz(dsa_mat_t* h, int64_t* out);
int32_t dsa_mat_size(dsa_mat_t* h, int64_t* m, int64_t* n);
int32_t dsa_mat_nbpartitions(dsa_mat_t* h, int32_t orientation, int64_t* out);
int32_t dsa_mat_info(dsa_mat_t* h, int32_t orientation, int64_t info[DSA_INFO_COUNT]);
/* parity probe: slot array + semaphores[id] (0 = nothing) + col_keys[id] with col_live[id] in {0,1} */
int32_t dsa_mat_export_layout(dsa_mat_t* h, int32_t orientation, int64_t* keys, double* vals,
                              uint8_t* occ, int64_t cap, int64_t* semaphores, int64_t* col_keys,
                              uint8_t* col_live, int64_t table_cap);
/* _even_rebalance!(pcsc, 1, capacity, nb_elements) of one orientation  src/pcsr.jl:88-97 */
int32_t dsa_mat_rebalance_root(dsa_mat_t* h, int32_t orientation);

/* ---- SpMV:  mat * v, transpose(mat) * v   src/operations.jl:14-60 -> _mul :107-135 ---- */
/* dense x (every index of x is a stored entry), dense y of length ny; rows never touched are 0.
 * transpose = 0: y = A x  (nx >= #cols used, ny = m) ; transpose = 1: y = A' x. */
int32_t dsa_mat_spmv_dense(dsa_mat_t* h, int32_t transpose, const double* x, int64_t nx,
                           double* y, int64_t ny);
/* sparse x given by its stored entries (xi ascending) ; result = touched rows only, ascending,
 * stored zeros kept: the shape of _mul_output(result, n)  src/operations.jl:11-12 */
int32_t dsa_mat_spmv_sparse(dsa_mat_t* h, int32_t transpose, const int64_t* xi, const double* xv,
                            int64_t nx, int64_t* yi, double* yv, int64_t cap, int64_t* n_out);
/* The same product in two steps, so that the caller allocates exactly what the result needs: _begin computes (result left with the
 * handle: packed in HBM, short ones also in pinned memory) and returns the number of touched rows, _fetch copies the pairs out
 * (DSA_ECAP when cap is too small: the result stays fetchable).  dsa_mat_spmv_sparse == _begin + _fetch.  One result per handle: the
 * next _begin / dsa_mat_spmv_sparse* call replaces it. */
int32_t dsa_mat_spmv_sparse_begin(dsa_mat_t* h, int32_t transpose, const int64_t* xi, const double* xv, int64_t nx, int64_t* n_out);
int32_t dsa_mat_spmv_sparse_fetch(dsa_mat_t* h, int64_t* yi, double* yv, int64_t cap, int64_t* n_out);
/* The sparse product with every operand in HBM, stream-ordered, no host wait: d_xi / d_xv = the nx stored entries of x (ascending
 * indices; not checked), d_yi / d_yv = cap entries for the touched rows (ascending), *d_count (device) = their number — pairs beyond
 * cap are dropped, the count still says how many there are (cap = size(m, 1 | 2) always suffices).  Enqueued on the stream of the
 * orientation that is walked (dsa_mat_set_stream / dsa_mat_sync).  With many stored entries (8 nx >= number of columns: gather over the
 * twin orientation) entries of x outside 1..n are ignored — column keys below 1 need the host entry point or fewer entries. */
int32_t dsa_mat_spmv_sparse_dev(dsa_mat_t* h, int32_t transpose, const int64_t* d_xi, const double* d_xv, int64_t nx,
                                int64_t* d_yi, double* d_yv, int64_t cap, int64_t* d_count);
/* same as dsa_mat_spmv_dense with x, y resident in HBM; asynchronous on the handle's stream.
 * algo: 0 = gather over the twin orientation (default), 1 = scatter over the reference's own
 * orientation with fp64 atomics (the literal _mul loop nest) */
int32_t dsa_mat_spmv_dense_dev(dsa_mat_t* h, int32_t transpose, int32_t algo, const double* d_x,
                               int64_t nx, double* d_y, int64_t ny);
/* ---- column-range shards.  No reference counterpart: the reference is single-process.  One PROCESS per GPU:
 * each process selects its device (dsa_set_device), builds ITS shard and runs the local SpMV; the single data-path collective —
 * the all-reduce (sum) of the partial y — is dsa_shard_allreduce_dev below (RCCL behind this ABI), or whatever the host layer has
 * (torch.distributed in bench.py / sharding.py).
 * dsa_shard_range: shard g of G owns the global column keys (col0, col0 + ncols].
 * dsa_shard_create_from_coo: the triples of the shard's range as an independent reference-layout matrix (own capacity, height,
 *   semaphores, column table; both orientations) with LOCAL column keys 1..ncols; size m x ncols.  n = global column count.
 * dsa_shard_spmv_dev: y_partial = A[:, range] * x[range]; d_x_local = the shard's slice of x (ncols doubles), d_y_partial = m
 *   doubles, both in HBM; asynchronous on the handle's stream (dsa_mat_set_stream / dsa_mat_sync). */
int32_t dsa_shard_range(int64_t n, int32_t nshards, int32_t shard, int64_t* col0, int64_t* ncols);
int32_t dsa_shard_create_from_coo(const int64_t* I, const int64_t* J, const double* V, int64_t nnz, int64_t m, int64_t n,
                                  int32_t nshards, int32_t shard, dsa_mat_t** out);
int32_t dsa_shard_spmv_dev(dsa_mat_t* shard, const double* d_x_local, int64_t nx, double* d_y_partial, int64_t ny);
/* ---- the one data-path collective: the sum of the partial y over the ranks, RCCL over xGMI, behind the ABI.
 * One process per GPU.  Rank 0 calls dsa_comm_unique_id and hands the 128 bytes to the other ranks by whatever means the host has
 * (MPI.Bcast from Julia, a file, torch.distributed); then EVERY rank calls dsa_comm_init (collective: ncclCommInitRank on the
 * current device).  librccl.so is bound at run time; a copy already mapped into the process is reused.
 * dsa_shard_allreduce_dev: in place, y <- sum over ranks (ncclAllReduce, ncclDouble, ncclSum), asynchronous on hip_stream.
 * dsa_shard_spmv_allreduce_dev: dsa_shard_spmv_dev + that all-reduce on the shard's stream = y = A x of the whole matrix on every
 * rank.  id == NULL with nranks == 1: a communicator without RCCL (single GPU); with an id a real single-rank RCCL communicator. */
#define DSA_COMM_ID_BYTES 128
typedef struct dsa_comm dsa_comm_t;
int32_t dsa_comm_unique_id(uint8_t id[DSA_COMM_ID_BYTES]);
int32_t dsa_comm_init(int32_t rank, int32_t nranks, const uint8_t id[DSA_COMM_ID_BYTES], dsa_comm_t** out);
int32_t dsa_comm_destroy(dsa_comm_t* comm);
int32_t dsa_comm_info(dsa_comm_t* comm, int32_t* rank, int32_t* nranks);
int32_t dsa_shard_allreduce_dev(dsa_comm_t* comm, double* d_y, int64_t m, void* hip_stream);
int32_t dsa_shard_spmv_allreduce_dev(dsa_mat_t* shard, dsa_comm_t* comm, const double* d_x_local, int64_t nx, double* d_y, int64_t ny);
/* Device-side invariant checker (the structural checks of the reference's test/utils.jl:68-113, runnable at full size):
 * report[0] occupied cells, [1] semaphore cells, [2] semaphore cells whose table entry does not point back, [3] key-order
 * violations, [4] bad table entries (not pointing at their semaphore / dead key / unsorted column keys), [5] occupancy bits at or
 * beyond the capacity, [6] 1 if report[0] != nb_elements or report[1] != live partitions.  A healthy structure has [2..6] == 0. */
int32_t dsa_vec_check(dsa_vec_t* h, int64_t report[8]);
int32_t dsa_mat_check(dsa_mat_t* h, int32_t orientation, int64_t report[8]);
/* stream control for *_dev entry points (hipStream_t passed as void*; NULL = legacy default stream) */
int32_t dsa_mat_set_stream(dsa_mat_t* h, void* hip_stream);
int32_t dsa_vec_set_stream(dsa_vec_t* h, void* hip_stream);
int32_t dsa_mat_sync(dsa_mat_t* h);
int32_t dsa_vec_sync(dsa_vec_t* h);
/* How the blocking entry points of a handle wait for the device.  No reference counterpart (the reference never waits for anything).
 * DSA_WAIT_SPIN (default): the calling thread polls a word in pinned memory that the last kernel of the launch writes — lowest
 * latency, one host core busy for the duration.  DSA_WAIT_BLOCK: the thread is parked in hipStreamSynchronize first — for hosts that
 * multiplex many tasks on few threads (a Julia process driving Coluna).  Results are identical; DSA_WAIT_POLICY=1 in the environment
 * makes DSA_WAIT_BLOCK the default of new handles. */
enum { DSA_WAIT_SPIN = 0, DSA_WAIT_BLOCK = 1 };
int32_t dsa_vec_set_wait_policy(dsa_vec_t* h, int32_t policy);
int32_t dsa_mat_set_wait_policy(dsa_mat_t* h, int32_t policy);
/* The library keeps freed HBM blocks (slot buffers, tables, build scratch) for reuse — up to DSA_POOL_MAX_MB (default 16384) of idle
 * memory per process.  dsa_pool_idle_bytes reports how much is idle right now, dsa_pool_trim releases idle blocks (largest first)
 * until at most keep_bytes remain: what a host that shares the card with another allocator calls before that one runs short. */
int32_t dsa_pool_idle_bytes(int64_t* bytes);
int32_t dsa_pool_trim(int64_t keep_bytes);
/* Release configuration.  The library has development switches (environment variables DSA_TIGHT, DSA_PARBATCH, DSA_MODEL3, ... that
 * select alternative — parity-tested, slower or instrumented — code paths for A/B runs and fault injection).  A release process IGNORES
 * all of them: they are honoured only when DSA_DEV=1 is set in the environment (or the library was built with -DDSA_DEV).  What stays
 * configurable without it is not a code-path switch: DSA_POOL_MAX_MB, DSA_RCCL_LIB, DSA_WAIT_POLICY, DSA_ROCTX.
 * dsa_dev_switches: the space-separated table of switch names (cap >= 1024 suffices) and whether this process honours them. */
int32_t dsa_dev_switches(char* buf, int64_t cap, int32_t* enabled);

/* ---- parity hooks and snapshots (no reference counterpart as entry points; the primitives they run are the reference's) ----
 * dsa_dbg_raw_*: ONE slot-array primitive of src/finds.jl / src/writes.jl / src/moves.jl executed by the DEVICE code on a
 * caller-supplied raw slot array (keys[i], vals[i], occ[i] in {0,1}, i < len; any length, any content) — the form the reference's
 * own unit tests use (test/unit/finds.jl:4-107, test/unit/writes.jl:5-70).  Arrays are uploaded, the kernel runs, the arrays (and
 * semaphores[], if given) are downloaded again.  engine selects which device implementation runs:
 *   DSA_DBG_ENGINE_BLOCK  the write sequencer's workgroup primitives (d_find / d_find_fast, d_insert_after + blk_shift_*, blk_purge,
 *                         blk_rebalance_small: windows <= 8192 slots)
 *   DSA_DBG_ENGINE_WAVE   the wave-level primitives of the batch-parallel rounds (pb_shift_*, pb_wave_rebalance: windows <= 2048 slots)
 *   DSA_DBG_ENGINE_GRID   the grid-wide rebalance kernel k_move2 (dsa_dbg_raw_rebalance only; windows of whole 64-slot words)
 * fast = 0: K-find replays the reference bisection probe for probe; fast = 1: its wave-parallel 64-ary form, which the write paths
 * use wherever the searched range is key-partitioned (only then are the two required to agree). */
enum { DSA_DBG_ENGINE_BLOCK = 0, DSA_DBG_ENGINE_WAVE = 1, DSA_DBG_ENGINE_GRID = 2 };
/* find(array, key, from, to)  src/finds.jl:29-57 -> (pos, elem) ; *has = 0 <=> elem === nothing */
int32_t dsa_dbg_raw_find(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t len, int64_t key, int64_t from, int64_t to,
                         int32_t engine, int32_t fast, int64_t* pos, int32_t* has, int64_t* fkey, double* fval);
/* insert!(array, key, value, from, to, semaphores)  src/writes.jl:14-43 ; sems may be NULL ; DSA_EFULL as the reference's error */
int32_t dsa_dbg_raw_insert(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t key, double value, int64_t from, int64_t to,
                           int64_t* sems, int64_t nsems, int32_t engine, int32_t fast, int64_t* pos, int32_t* is_new);
/* delete!(array, key, from, to)  src/writes.jl:57-68 */
int32_t dsa_dbg_raw_delete(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t key, int64_t from, int64_t to,
                           int32_t engine, int32_t fast, int64_t* pos, int32_t* deleted);
/* purge!(array, from, to)  src/writes.jl:80-91 -> (mid, number of cells deleted) */
int32_t dsa_dbg_raw_purge(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t from, int64_t to, int64_t* mid, int64_t* nb);
/* pack! + spread! of the window [ws, we] (its cell count is taken from occ)  src/moves.jl:94-171 ; with sems: the semaphore-aware
 * spread! (:142-171).  Windows lie inside one occupancy word or consist of whole words, like every window the density scan yields. */
int32_t dsa_dbg_raw_rebalance(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t ws, int64_t we, int64_t* sems, int64_t nsems,
                              int32_t engine);
/* Handles restored from an exported layout (the inverse of dsa_vec_export_layout / dsa_pcsc_export_layout: snapshot / restore, and
 * the way a test puts a structure into an arbitrary state).  capacity and segment_capacity are powers of two (segment_capacity is
 * state, not a function of the capacity: SURVEY App. A.1); nb_segments, height and the density thresholds follow as in
 * src/pma.jl:42-49,143-161; nb_elements = number of occupied slots; semaphores[id] = slot of the cell (0, id) or 0 (tombstone). */
int32_t dsa_vec_import_layout(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t capacity, int64_t segment_capacity,
                              int64_t len, dsa_vec_t** out);
int32_t dsa_pcsc_import_layout(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t capacity, int64_t segment_capacity,
                               const int64_t* semaphores, int64_t table_len, dsa_pcsc_t** out);

#ifdef __cplusplus
}
#endif
#endif /* DSA_H */
