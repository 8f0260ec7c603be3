// csrc/dsa_dev.h — shared declarations of the gfx950 PMA / PCSR engine.
//
// Physical layout in HBM (ours; the LOGICAL slot array it encodes is bit-identical to the
// reference's Elements{K,T}, src/DynamicSparseArrays.jl:18):
//   int64  keys[cap_alloc]        8 B / slot   (coalesced streams for SpMV / rebalance)
//   double vals[cap_alloc]        8 B / slot
//   uint64 occ [cap_alloc / 64]   1 bit / slot — bit i of word w <=> slot 64*w+i (0-based) holds a
//                                 tuple; one word is exactly one wave64 ballot.  Bits >= capacity are 0.
//   int64  sems[table_cap]        semaphores[id] = 1-based slot of partition id's semaphore, 0 = nothing
//   int64  col_keys[table_cap] + uint8 col_live[table_cap]   (MappedPackedCSC.col_keys, src/pcsr.jl:16-19)
// Positions in device code are 1-based like the reference; array index = pos - 1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <mutex>

namespace dsa {

constexpr int WAVE = 64;
constexpr int64_t SEM_KEY = 0;            // semaphore_key(::Type{<:Integer})  src/pcsr.jl:23
constexpr int MAX_LEVELS = 48;
// partition-table entries a batch may leave unmerged at the END of the tables (Ctl::n_pending): the one limit the sequencer's LDS
// list (sequencer.hip), the batch-parallel rounds (parbatch.hip) and the grid-wide merge (tables.hip) share
constexpr int TABLE_PEND_MAX = 1024;
// slot ranges up to this size are packed by ONE launch of one workgroup (sequencer.hip: k_view_small): column views, slices, small vectors
constexpr int64_t VIEW_SMALL_SLOTS = 65536;

// ---- physical key storage ---------------------------------------------------------------------------
// K = Int64 at the API; in HBM a slot array keeps its keys in 32 bits as long as every key ever written fits Int32
// (row / column indices practically always do) and is widened once, by the host, before the first key that does not
// (`wide`).  25 % fewer bytes for every kernel that streams slots (rebalance, SpMV, K-permute, K-build).  Kernels
// receive the array by value; keys[i] reads / writes through a proxy, the streaming kernels use the typed loads below.
struct KeyArr {
    void* p;
    int32_t wide;          // 0: int32_t keys, 1: int64_t keys
    int32_t pad_;
#if defined(__HIPCC__)
    struct Ref {
        void* p; int32_t wide; int64_t i;
        __device__ __forceinline__ operator int64_t() const {
            return wide ? static_cast<const int64_t*>(p)[i] : (int64_t) static_cast<const int32_t*>(p)[i];
        }
        __device__ __forceinline__ void operator=(int64_t k) const {
            if (wide) static_cast<int64_t*>(p)[i] = k; else static_cast<int32_t*>(p)[i] = (int32_t)k;
        }
        __device__ __forceinline__ void operator=(const Ref& o) const { *this = (int64_t)o; }
    };
    __device__ __forceinline__ Ref operator[](int64_t i) const { return Ref{p, wide, i}; }
    __device__ __forceinline__ int64_t ld(int64_t i) const { return (int64_t)(*this)[i]; }
    __device__ __forceinline__ int64_t ld_nt(int64_t i) const {            // streamed once
        return wide ? __builtin_nontemporal_load(static_cast<const int64_t*>(p) + i)
                    : (int64_t)__builtin_nontemporal_load(static_cast<const int32_t*>(p) + i);
    }
    __device__ __forceinline__ int64_t ld_agent(int64_t i) const {         // L2-served (cells just written by this launch)
        return wide ? __hip_atomic_load(static_cast<const int64_t*>(p) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                    : (int64_t)__hip_atomic_load(static_cast<const int32_t*>(p) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // two adjacent keys (i even), non-temporal: one 16-byte / 8-byte store
    __device__ __forceinline__ void st2_nt(int64_t i, int64_t k0, int64_t k1) const {
        if (wide) {
            typedef long long ll2v __attribute__((ext_vector_type(2)));
            ll2v kv; kv.x = k0; kv.y = k1;
            __builtin_nontemporal_store(kv, reinterpret_cast<ll2v*>(static_cast<int64_t*>(p) + i));
        } else {
            typedef int i2v __attribute__((ext_vector_type(2)));
            i2v kv; kv.x = (int)k0; kv.y = (int)k1;
            __builtin_nontemporal_store(kv, reinterpret_cast<i2v*>(static_cast<int32_t*>(p) + i));
        }
    }
#endif
};
static inline bool key_fits32(int64_t k) { return k >= INT32_MIN && k <= INT32_MAX; }

// ---- control block shared by host and the sequencer kernel (one per PMA) ----------------------
enum SeqStatus : int32_t {
    SEQ_DONE = 0,
    SEQ_Y_REBALANCE = 1,   // host must run the big pack+spread on [y_ws, y_we] with y_m cells
    SEQ_Y_EXTEND = 2,      // root density > t  -> _extend!  (src/pma.jl:143-151) then root rebalance
    SEQ_Y_SHRINK = 3,      // root density < p  -> pack + _shrink! (src/pma.jl:135-139,153-161)
    SEQ_Y_TABLE_GROW = 4,  // semaphore / col_keys tables are full; re-run the op after growing
    SEQ_ERROR = 5,
    SEQ_Y_APPEND_RUN = 6   // ops [y_ws, y_ws+y_m) are ascending appends behind the y_we cells: the host saves the bitmap, runs
                           // k_append_run (bitmap-only replay) and K-permute (one move per cell), then relaunches
};

struct Ctl {
    // PackedMemoryArray scalars (src/pma.jl:8-24)
    int64_t capacity, segment_capacity, nb_segments, nb_elements, height;
    // integer form of the density thresholds: level h (window W_h = seg*2^h) is accepted iff
    // lo[h] <= count <= hi[h]  <=>  p_0+p_d*h <= count/W_h <= t_0+t_d*h in Float64 (W_h is a power
    // of two, so count/W_h and p*W_h are exact).  Computed on the host in plain IEEE doubles.
    int64_t lo[MAX_LEVELS], hi[MAX_LEVELS];
    // PackedCSC tables
    int64_t nb_partitions, table_len, table_cap;
    // batch progress / yield mailbox
    int64_t next_op;
    int32_t status, err;
    int64_t y_ws, y_we, y_m;
    int64_t err_op;
    // instrumentation
    int64_t stat_window_slots, stat_rebalances, stat_extends, stat_shrinks, stat_small_rebalances;
    int64_t no_run_at;     // op index that must take the normal path (an append run made no progress there), or -1
    int64_t n_pending;     // table entries [table_len - n_pending, table_len) were created by the running batch at the END of the tables
                           // (arrival order, not key order; <= 1024); merged into key order by tables.hip; 0 at every API boundary
    int64_t prof[16];      // dev profile of the sequencer (shader cycles): table lookup, new partition, element write, merges; counts
    int64_t dbg[6];        // append-run profile of the last run: slow ops, ticks (100 MHz) in setup / fast loop / slow path, blocks loaded
    // vector length n (src/vector.jl:2) is host-only
};

enum OpKind : int32_t {
    OP_VEC_SET = 0,        // setindex!(pma, v, key)                    src/pma.jl:196-213       a=key
    OP_PCSC_SET = 1,       // setindex!(pcsc, v, key, partition)        src/pcsr.jl:294-310      a=key b=partition
    OP_MPCSC_SET = 2,      // setindex!(mpcsc, v, row, col)             src/pcsr.jl:341-351      a=row b=col
    OP_DELETE_PARTITION = 3,   // deletepartition!(pcsc, p)             src/pcsr.jl:188-204      b=partition
    OP_MPCSC_DELETECOLUMN = 4  // deletecolumn!(mpcsc, col)             src/pcsr.jl:206-212      b=col
};

struct Op {
    int64_t a, b;
    double v;
    int32_t kind, pad;
};

// error codes (include/dsa.h)
enum : int32_t { E_OK = 0, E_ARG = 1, E_BOUNDS = 2, E_DELETED = 3, E_FULL = 4, E_MODE = 5, E_ASSERT = 6, E_HIP = 7, E_CAP = 8, E_KEY = 9 };

#if defined(__HIPCC__)
// ---- bit helpers --------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint64_t mask_lt(int i) { return i <= 0 ? 0ull : (i >= 64 ? ~0ull : ((1ull << i) - 1ull)); }   // bits [0, i)
__device__ __forceinline__ bool occ_test(const uint64_t* occ, int64_t pos) {
    const int64_t i = pos - 1;
    return (occ[i >> 6] >> (i & 63)) & 1ull;
}
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }

// ---- spread! geometry (src/moves.jl:120-171) ------------------------------------------------------
// After spread! of m cells over a W-slot window, the empty slots are exactly the offsets
//   D(k) = floor(fl(k * fl(W / E))),  k = 1..E,  E = W - m         (1-based offsets in the window)
// evaluated in IEEE Float64; the m cells fill the other offsets in order.
struct SpreadGeom {
    int64_t W, E;
    double f;       // fl(W / E)   (unused when E == 0)
    double inv_f;   // E / W, only a starting guess
};
__device__ __host__ inline SpreadGeom make_geom(int64_t W, int64_t m) {
    SpreadGeom g;
    g.W = W; g.E = W - m;
    g.f = g.E > 0 ? (double)W / (double)g.E : 0.0;
    g.inv_f = (double)g.E / (double)W;
    return g;
}
// Window sizes are < 2^31 slots (checked on the host), so offsets / gap indices fit int32 and the
// int <-> double conversions are single instructions (v_cvt_f64_i32 / v_cvt_i32_f64).
__device__ __forceinline__ int gap_D(const SpreadGeom& g, int k) {
    return (int)floor(__dmul_rn((double)k, g.f));
}
// number of empty offsets <= q, q in [0, W].  D(k) <= q  <=>  k*f < q+1, so the count is within one of
// floor((q+1)*E/W); the two fix-up loops make it exact for the Float64 D and almost never iterate twice.
__device__ __forceinline__ int gaps_le(const SpreadGeom& g, int q) {
    const int E = (int)g.E;
    if (E <= 0) return 0;
    int k = (int)((double)(q + 1) * g.inv_f);
    if (k > E) k = E;
    if (k < 0) k = 0;
#pragma clang loop vectorize(disable) unroll(disable)
    while (k < E && gap_D(g, k + 1) <= q) ++k;
#pragma clang loop vectorize(disable) unroll(disable)
    while (k > 0 && gap_D(g, k) > q) --k;
    return k;
}
// offset q in [1, W]: returns true if q is a gap; otherwise *rank = 1-based rank of the cell landing on q
__device__ __forceinline__ bool slot_is_gap(const SpreadGeom& g, int q, int* rank) {
    const int k = gaps_le(g, q);
    if (k > 0 && gap_D(g, k) == q) return true;
    *rank = q - k;
    return false;
}
// Two adjacent offsets q, q+1 at once with two evaluations of D in the common case:
// *k = number of gaps <= q ; gap0 / gap1 = whether q / q+1 are gaps.
__device__ __forceinline__ void gap_pair(const SpreadGeom& g, int q, int* k_out, bool* gap0, bool* gap1) {
    const int E = (int)g.E;
    if (E <= 0) { *k_out = 0; *gap0 = false; *gap1 = false; return; }
    int k = (int)((double)(q + 1) * g.inv_f);
    if (k > E) k = E;
    if (k < 0) k = 0;
    int d0 = k > 0 ? gap_D(g, k) : 0;                 // D(k)   (0 when k == 0: smaller than every offset)
    int d1 = k < E ? gap_D(g, k + 1) : 0x7fffffff;    // D(k+1)
#pragma clang loop vectorize(disable) unroll(disable)
    while (d1 <= q) { ++k; d0 = d1; d1 = k < E ? gap_D(g, k + 1) : 0x7fffffff; }
#pragma clang loop vectorize(disable) unroll(disable)
    while (k > 0 && d0 > q) { --k; d1 = d0; d0 = k > 0 ? gap_D(g, k) : 0; }
    *k_out = k;
    *gap0 = (k > 0 && d0 == q);
    *gap1 = (d1 == q + 1);
}
// occupancy of window offsets 64t+1 .. 64t+64 after spread! (bit b <-> offset 64t+1+b)
__device__ __forceinline__ uint64_t spread_word_bits(const SpreadGeom& g, int t) {
    uint64_t bits = ~0ull;
    const int E = (int)g.E;
    const int lo = 64 * t, hi = lo + 64;
    int k = gaps_le(g, lo);
#pragma clang loop vectorize(disable) unroll(disable)
    for (++k; k <= E; ++k) {
        const int d = gap_D(g, k);
        if (d > hi) break;
        bits &= ~(1ull << (d - lo - 1));
    }
    return bits;
}
// 1-based offset of the last cell after spread! of m >= 1 cells over W slots
__device__ __forceinline__ int64_t spread_last_cell(const SpreadGeom& g) {
    int q = (int)g.W, k = (int)g.E;
    while (k > 0 && gap_D(g, k) == q) { --q; --k; }
    return q;
}
// bit-interleave: bit i of x -> bit 2i
__device__ __forceinline__ uint64_t spread_bits32(uint32_t x) {
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
#endif  // __HIPCC__

void set_last_error(const char* msg);
// Development switches (A/B runs, fault injection, coverage of alternative code paths).  A RELEASE process ignores every one of them:
// dev_env() returns nullptr unless DSA_DEV=1 is set in the environment, so that a parity-critical path cannot be selected by whatever
// the host application happens to inherit.  The names are one table in dsa_host.hip (dsa_dev_switches lists it); a name that is
// not in the table aborts in the debug build.  Configuration that is NOT a development switch keeps plain getenv: DSA_POOL_MAX_MB,
// DSA_RCCL_LIB, DSA_WAIT_POLICY, DSA_ROCTX.
const char* dev_env(const char* name);      // text behind dsa_last_error_message() for entry points outside dsa_host.hip

// one-time kernel attribute setup (hipFuncSetAttribute is per device): thread-safe — the two orientations of a matrix are driven
// from two host threads (dsa_host.hip: mat_apply_sets) — and repeated for every device a process uses
struct PerDeviceOnce {
    static constexpr int MAX_DEV = 32;
    std::once_flag flag[MAX_DEV];
    hipError_t err[MAX_DEV];
    template <typename F> hipError_t run(F fn) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEV) return fn();
        std::call_once(flag[d], [&] { err[d] = fn(); });
        return err[d];
    }
};

// ---- host-side launch wrappers (defined in the .hip files) ----------------------------------------
struct RebalanceWork {   // scratch owned by a PMA for the big pack+spread
    uint32_t* tile_cnt;    // one counter per 4096-slot source tile
    uint32_t* tile_off;    // exclusive prefix
    int64_t tiles_cap;
    unsigned long long* status = nullptr;   // k_move2: one look-back status word per 2048-slot source tile (zeroed at allocation)
    int64_t status_cap = 0;
    unsigned long long gen = 0;              // generation of the last launch: stale words of earlier launches are "not ready"
};

// gathers the m occupied cells of src[src_ws..src_we] (in order) and spreads them over
// dst[dst_ws..dst_we]; writes every dst slot, the dst occupancy words and, if sems != nullptr,
// semaphores[id] for every cell with key == 0.  src_packed: the cells are src slots src_ws..src_ws+m-1.
hipError_t launch_rebalance(KeyArr src_keys, const double* src_vals, const uint64_t* src_occ,
                            int64_t src_ws, int64_t src_we, bool src_packed,
                            KeyArr dst_keys, double* dst_vals, uint64_t* dst_occ,
                            int64_t dst_ws, int64_t dst_we, int64_t m, int64_t* sems,
                            RebalanceWork* work, hipStream_t stream);
// bench hook: the m cells of slots 0..m-1 of (src) -> the LAST m slots of a cap-slot destination, bitmap included
hipError_t launch_pack_right(KeyArr src_keys, const double* src_vals, int64_t m, KeyArr dst_keys, double* dst_vals, uint64_t* dst_occ,
                             int64_t cap, hipStream_t stream);
// K-permute: order-preserving move of the n0 cells of (src, src_occ) followed by the cells of ops[i0..] to the set bits of
// dst_occ (already final), writing dst keys / vals and, if sems != nullptr, the semaphore table
hipError_t launch_permute(KeyArr src_keys, const double* src_vals, const uint64_t* src_occ, int64_t src_cap,
                          KeyArr dst_keys, double* dst_vals, const uint64_t* dst_occ, int64_t dst_cap, int64_t n0,
                          const Op* ops, int64_t i0, int64_t* sems, RebalanceWork* wsrc, RebalanceWork* wdst, hipStream_t stream);
// K-pack: occupied cells of slots [from, to] (1-based, inclusive), in slot order, to dense device buffers of capacity out_cap;
// *count (host) receives the number of cells.  Synchronises the stream once (count needed to size the copy-out).
hipError_t launch_compact_range(KeyArr keys, const double* vals, const uint64_t* occ, int64_t from, int64_t to,
                                KeyArr out_keys, double* out_vals, int64_t out_cap, RebalanceWork* work, int64_t* count,
                                hipStream_t stream);
// sparse-x product (sparsex.hip): accumulate driven by x's stored entries, touched-row bitmap of a pattern pass, count + emit
hipError_t launch_spx_accum(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, const int64_t* sems, const int64_t* col_keys,
                            const uint8_t* col_live, int64_t table_len, const int64_t* xi, const double* xv, int64_t nx, double* acc,
                            uint64_t* bm, int64_t ny, hipStream_t stream);
hipError_t launch_spx_pattern_bits(const double* pattern, int64_t ny, uint64_t* bm, hipStream_t stream);
hipError_t launch_spx_finish(uint64_t* bm, int64_t ny, uint32_t* tile_cnt, uint32_t* tile_off, unsigned int* ticket, double* src, int clear_src,
                             int64_t* out_i, double* out_v, int64_t cap, int64_t* d_count, long long* host, int64_t pin_cells,
                             unsigned long long seq, hipStream_t stream);
hipError_t launch_scatter_x(const int64_t* xi, const double* xv, int64_t nx, double* xd, double* xf, int64_t nxd, hipStream_t stream);
// whole-vector operations on packed streams (rebalance.hip)
hipError_t launch_packed_equal(KeyArr ka, const double* va, KeyArr kb, const double* vb, int64_t n, int32_t* differ, hipStream_t stream);
hipError_t launch_merge_axpby(KeyArr ka, const double* va, int64_t na, double alpha, KeyArr kb, const double* vb, int64_t nb,
                              double beta, int64_t* mk, double* mv, uint64_t* keep, hipStream_t stream);
hipError_t launch_widen_keys(const void* src32, void* dst64, int64_t n, hipStream_t stream);
// a structure made of fresh (uninitialised) blocks in ONE launch: both occupancy bitmaps and the status table of the grid rebalance
// zeroed, the control block written from the argument (no copy command)
hipError_t launch_store_ctl(Ctl* d_ctl, const Ctl& ctl, hipStream_t stream);
hipError_t launch_init_fresh(uint64_t* occ0, uint64_t* occ1, int64_t occ_words, unsigned long long* status, int64_t status_words,
                             Ctl* d_ctl, const Ctl& ctl, hipStream_t stream);
// clears occupancy bits of slots [from, to] (1-based, inclusive); from/to word-aligned or inside one word
hipError_t launch_clear_occ(uint64_t* occ, int64_t from, int64_t to, hipStream_t stream);

// ---- caching allocator for HBM blocks (pool.hip): the scratch of a bulk build and the slot buffers of a structure come from
// here; a block handed back must not be referenced by work still in flight
hipError_t pool_alloc(void** out, size_t bytes);
void pool_free(void* p);
void pool_trim(size_t keep_bytes);
size_t pool_idle_bytes();
hipError_t pinned_alloc(void** out, size_t bytes);      // pinned host blocks, kept by size class
void pinned_free(void* p);
hipError_t stream_get(hipStream_t* out);                // a non-blocking stream of the current device, kept when a handle dies
void stream_put(hipStream_t s, int device);             // (the caller has synchronised it)

// ---- K-build (build.hip): device bulk constructor of one orientation ---------------------------------------------
struct BuildScratch {
    int64_t n = 0;
    void* base = nullptr; hipStream_t stream = nullptr;      // the block (pool.hip) the device arrays below are carved from
    void* base_val = nullptr;                                // ... except val[0..1], allocated only when the values travel with the sorted words (ibits == 0)
    // composite path: comp = (partition - pmin) << kbits | (key - kmin), sorted with its value (build.hip)
    void* d_ctl = nullptr; void* h_ctl = nullptr;            // BuildCtl on the device / its pinned mirror
    uint32_t* ghist = nullptr; uint32_t* hist = nullptr; uint32_t* cnt_c = nullptr; uint32_t* cnt_p = nullptr;
    uint64_t* comp[2] = {nullptr, nullptr}; double* val[2] = {nullptr, nullptr};
    void* queue = nullptr;
    int64_t kmin = 0, pmin = 0; int kbits = 0, pbits = 0, sorted = 0; const double* vsorted = nullptr;
    int ibits = 0;                                           // > 0: the sorted words are composite << ibits | input index, values are gathered from the caller's array
    // general path (composite wider than 64 bits): sorted input index, keys and partitions, flags and their inclusive prefix sums
    bool wide_path = false;
    uint32_t *idx2 = nullptr, *fpart = nullptr, *fcell = nullptr, *spart = nullptr, *scell = nullptr;
    int64_t *p2 = nullptr, *k2 = nullptr;
};
// bounds of the values of a key array (closed; need not be tight): they fix how many bits of the composite are sorted.  The host has
// seen every key it uploads (it scans them for their storage width anyway); unknown() ranges cost one min / max pass on the device.
struct KeyRange { int64_t lo = 0, hi = -1; bool known() const { return hi >= lo; } };
hipError_t device_key_scan(const int64_t* d_a, const int64_t* d_b, int64_t n, KeyRange* ra, KeyRange* rb, bool* a_zero, bool* b_zero,
                           hipStream_t stream);
// the same for a stream that arrives in pieces: folds n more entries into d_acc = {a min, a max, b min, b max, zero flags} (5 x int64 in
// HBM, initialised to {INT64_MAX, INT64_MIN, INT64_MAX, INT64_MIN, 0}); stream-ordered, no host wait
hipError_t launch_key_scan_acc(const int64_t* d_a, const int64_t* d_b, int64_t n, long long* d_acc, hipStream_t stream);
// phase 1: sort by (partition, key, input order), flags, counts; counts[0] = distinct cells, counts[1] = partitions
// while_sorting (or nullptr): called once the sort kernels and the copy of the counts are in flight, before the wait for them
hipError_t build_prepare(const int64_t* d_part, const int64_t* d_key, const double* d_val, int64_t nnz, KeyRange part_range, KeyRange key_range,
                         BuildScratch& s, int64_t counts[2], hipStream_t stream, const std::function<void()>* while_sorting = nullptr);
// phase 2: emit the ordered cell stream [sem(0,id), entries...] (counts[0]+counts[1] cells) and the partition keys
// mode 0: mapped partitions (semaphores + partition keys) ; 1: plain vector (d_part was nullptr) ; 2: explicit partition
// ids 1..nparts_explicit in d_part (PackedCSC: empty partitions keep their semaphore)
// wait_and_free = false: the kernels are only enqueued; the caller waits for the stream itself and releases the scratch with build_abort
hipError_t build_emit(const double* d_val, int32_t combine, BuildScratch& s, KeyArr out_keys, double* out_vals,
                      int64_t* part_keys, int mode, int64_t nparts_explicit, hipStream_t stream, bool wait_and_free = true,
                      uint64_t* der_comp = nullptr, double* der_val = nullptr);
// the twin orientation of a matrix from the cells its sibling's emit leaves behind (der_comp / der_val above == comp[0] / val[0] of the
// twin's scratch): build_derived_alloc before the sibling's emit, build_derived_sort behind it, then build_emit on the twin's scratch
hipError_t build_derived_alloc(BuildScratch& s, int64_t n, int kbits, int pbits, int64_t kmin, int64_t pmin, hipStream_t stream);
hipError_t build_derived_sort(BuildScratch& s, int64_t counts[2], hipStream_t stream, const std::function<void()>* while_sorting = nullptr);
void build_abort(BuildScratch& s);
// a vector of up to 1024 entries by one launch; io = pinned landing area (see k_build_small_vec)
hipError_t launch_build_small_vec(int64_t* io, int n, int cap, int32_t combine, KeyArr out_k, double* out_v, unsigned long long seq, hipStream_t stream);

// n_avail >= n_ops: ops resident behind the chunk (an append run may consume them); run_ok enables append-run detection
// breaks: the bitmap of launch_op_breaks for these ops (run detection reads it instead of the ops), or nullptr
hipError_t launch_sequencer(KeyArr keys, double* vals, uint64_t* occ, int64_t* sems, int64_t* col_keys,
                            uint8_t* col_live, Ctl* ctl, const Op* ops, int64_t n_ops, int64_t n_avail, bool run_ok, const uint64_t* breaks,
                            Ctl* host_ctl, unsigned long long* host_seq, unsigned int seq,
                            hipStream_t stream);
// one bit per op: op j does not continue an append run from op j-1 (mode 0: vector ops, 1: MappedPackedCSC ops); n/64 + 1 words
hipError_t launch_op_breaks(const Op* ops, int64_t n, int mode, uint64_t* breaks, hipStream_t stream);
// the op array of a batch from its columns in HBM (a, b or nullptr, v): op k = (a[k], b[k] or 0, v[k], kind)
hipError_t launch_make_ops(const int64_t* a, const int64_t* b, const double* v, int32_t kind, int64_t n, Op* ops, hipStream_t stream);
// flags / d_T: cell types and cell count written by k_run_expand (MappedPackedCSC runs), nullptr for a vector run
// saved_memo: append_run_memo_bytes() of zero-initialised device memory owned by the handle (the replay's memo survives in it from
// run to run while the array's geometry stays the same), or nullptr
size_t append_run_memo_bytes();
// m3_out: the 8 words k_append_model3 wrote for this run (cells it placed, status), or nullptr
hipError_t launch_append_run(uint64_t* occ, Ctl* ctl, int64_t i0, int64_t R, const uint64_t* flags, const int64_t* d_T,
                             uint64_t* saved_memo, const int64_t* m3_out, hipStream_t stream);
// count-only replay of an append run (appendmodel.hip), launched in front of launch_append_run on the same bitmap / control block.
// out = 8 x int64 device words: [0] cells placed [1] 0 not eligible (nothing changed) / 1 ran / 2 ran and stopped in front of an op
// that needs _extend! [2] reason when not eligible [3..7] dev counters (events above the tables, table levels, ticks per phase)
hipError_t launch_append_model3(uint64_t* occ, Ctl* ctl, int64_t R, const uint64_t* flags, const int64_t* d_T, int64_t* out, hipStream_t stream);
// typed replay of an append run WITH semaphore cells on 8-slot segments (appendmodel.hip: k_append_model5; a matrix grown from the empty
// one — BASELINE config 5); same hand-over words as launch_append_model3; consumes whole leaf epochs, the rest is launch_append_run's
hipError_t launch_append_model5(uint64_t* occ, Ctl* ctl, int64_t R, const uint64_t* flags, const int64_t* d_T, int64_t* out, hipStream_t stream);
hipError_t launch_run_expand(const Op* ops, int64_t i0, int64_t R, const Ctl* ctl, int64_t* col_keys, uint8_t* col_live, Op* cells,
                             uint64_t* flags, int64_t* out, hipStream_t stream);

// ---- grid-wide merge of the pending partition-table entries (tables.hip) -------------------------------------------------
struct TableMerge {
    int64_t* sems2; int64_t* keys2;               // scratch tables, table_cap entries each
    int64_t* pkey; int64_t* pdst; int64_t* psem;  // pending entries by key rank: key, destination index, semaphore slot (1024 each)
    int64_t* hdr;                                 // [0] K (0: nothing to do) [1] first table index that moves [2] fault flag
};
hipError_t launch_table_merge(int64_t* sems, int64_t* col_keys, uint8_t* col_live, double* vals, Ctl* ctl, TableMerge tm, int64_t table_cap,
                              hipStream_t stream);

// ---- batch-parallel writes of a vector's PMA (parbatch.hip) ------------------------------------------------------------
struct Plan {
    int64_t lo, hi;        // footprint: slots the op reads for its decisions or modifies (lo > hi: empty)
    int64_t pos, aux;      // predecessor / found position ; shift target (next / previous empty slot)
    int64_t ws, we;        // window accepted by the density scan after the op
    int32_t count;         // cells of that window after the op
    int32_t action;
};
// device-resident driver state of the batch-parallel rounds: the host enqueues several rounds back to back and only then
// synchronises to read it
struct RoundState {
    int64_t cursor;        // next op of the batch
    int64_t limit;         // one past the last op
    int32_t G;             // ops planned per round (adapted on the device to 2x the last prefix, 64..1024)
    int32_t d;             // prefix length decided by the resolve step (k_plan's last workgroup) for the round in flight
    int32_t stop;          // 0 running, 1 short prefix at `cursor` (sequencer must take over), 2 batch finished
    int32_t min_prefix, G_next, pad;   // pad: fault flag raised by k_apply (an op left its planned footprint: cannot happen, checked by the host)
    uint32_t ticket;       // workgroups of k_plan that have finished (the last one resolves the round; 0 between launches)
    int32_t ema;           // running average of the prefix lengths x16 (carried from burst to burst by the host): a short prefix only stops
                           // the rounds while the recent ones were short too (appends, one hot key), not for one unlucky collision
    int64_t rounds, par_ops;
    int64_t why[8];        // dev: what cut the prefixes (index = Plan::count of the first BARRIER op; 7 = a conflict)
    int32_t tight;         // tight footprints (parbatch.hip): bit 0 leaf-accepted inserts / deletes, bit 1 leaf-accepted new columns (dev knob DSA_TIGHT, default 3)
    int32_t seq;           // number of the burst (set by the host): k_publish hands it back with the state
    // ---- run-ahead (round 6): a round applies every op of its window that conflicts with no EARLIER op and lies outside the sealed zones
    // of the deferred ones, not only the prefix in front of the first conflict.  The window of a round is the PENDING list (ops deferred
    // by earlier rounds, ascending op index, DevBufs::pend[cur]) followed by fresh ops from `cursor`.  (cursor, np, cur) describe the round
    // whose k_apply runs; (cursor_n, np_n, cur_n) the next one (written by the resolve step, committed by the next round's).
    int64_t cursor_n;      // first fresh op of the next round
    int64_t pend0;         // op index of position 0 of the next window (what the sequencer takes when that op cannot be planned)
    int32_t np, np_n;      // pending ops in front of the window: of the round in flight / of the next one
    int32_t cur, cur_n;    // which half of DevBufs::pend holds that list
    int32_t run_ahead;     // 0: prefix rule (the host: batches whose ops can fail, A/B)   1: run-ahead
    int32_t drain;         // 1: the window is the pending list alone (the host is about to hand over to the sequencer / the local rounds)
    int64_t deferred;      // instrumentation: ops deferred behind a conflict inside the applied range (sum over the rounds)
};
// a pending op and the zone it may still touch (the intersection of the sealed zones it was deferred with; lo = 0: no constraint)
struct PendOp { int64_t op; int32_t zlo, zhi; };
// where a burst leaves its result for the host: the pinned mirrors of the round state and of the control block, and the word the
// host polls (the burst number) — written by k_publish, the last kernel of a burst, with system-scope stores: no copy commands, no
// stream synchronisation on the host side (nullptr: the host copies and synchronises itself)
struct BurstPublish { RoundState* host_rs; Ctl* host_ctl; unsigned long long* host_seq; };
// footprint-check build (parbatch.hip, -DDSA_FP_CHECK): bits of RoundState::tight that select the check, and the bytes per op the
// plan array carries behind the plans for the recorded read / touch sets and the final footprints
constexpr int FP_MODE_SETS = 0x100, FP_MODE_SHADOW = 0x200;
constexpr size_t FP_BYTES_PER_OP = 160;
constexpr int ROUND_GMAX = 1024;       // ops planned per round at most (parbatch.hip: one wave each; the host sizes the plan array and the pending lists).
                                       // 2048 was measured in round 6: the resolve step costs per planned op, a round of twice the size takes twice as long
struct BurstGraph {        // cached hipGraph of one burst of rounds (host-side)
    hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; hipStream_t stream = nullptr;
    const void* key[12] = {};
    bool disabled = false;              // capture failed on `failed_on` (e.g. the legacy null stream): eager launches until the stream changes
    hipStream_t failed_on = nullptr;
};
// the arrays the rounds work on, read by the kernels from device memory: a root rebalance swaps the slot buffers and the tables
// grow, but the launch arguments — and with them the cached graph — stay the same (re-instantiating the graphs after every
// _extend! cost ~1 ms a time while an array was growing)
struct DevBufs { void* keys; double* vals; uint64_t* occ; int64_t* sems; int64_t* col_keys; uint8_t* col_live; int32_t wide, pad; PendOp* pend; /* 2 x ROUND_GMAX */ };
hipError_t launch_burst(const DevBufs* bufs, Ctl* ctl, const Op* ops, RoundState* rs, Plan* plans, int rounds, BurstGraph* cache,
                        BurstPublish pub, hipStream_t stream);
void burst_graph_destroy(BurstGraph* cache);
// the same rounds by one persistent workgroup (phases with short conflict-free prefixes); leaves with RoundState::stop = 0 (max_rounds
// used up), 1 (the op at the cursor needs the sequencer), 2 (batch finished) or 3 (full prefixes: back to the grid rounds)
hipError_t launch_local_rounds(const DevBufs* bufs, Ctl* ctl, const Op* ops, RoundState* rs, int max_rounds, BurstPublish pub, hipStream_t stream);

// batched read-only lookups.  mode 0: getindex(pma, key) ; 1: getindex(pcsc, key, partition) ;
// 2: getindex(mpcsc, row, col).  err_out: first error code (0 if none)
hipError_t launch_get_batch(int mode, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                            const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live, int64_t table_len,
                            const int64_t* qa, const int64_t* qb, int64_t n, double* out, int32_t* err_out,
                            hipStream_t stream);
// up to 64 lookups through a pinned landing area (no copy commands; see k_get_small)
hipError_t launch_get_small(int mode, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, const int64_t* sems,
                            const int64_t* col_keys, const uint8_t* col_live, int64_t table_len, int64_t* io, int n, unsigned long long seq,
                            hipStream_t stream);
// parity hooks (include/dsa.h: dsa_dbg_raw_*): ONE slot-array primitive on a raw slot array of `len` slots; out = 6 x int64 device
// scratch {error, position, flag, found key, found value bits, cells purged}.  _block: the sequencer's workgroup primitives
// (sequencer.hip), _wave: the wave-level primitives of the batch-parallel rounds (parbatch.hip).  *_FAST: K-find in its
// wave-parallel 64-ary form (d_find_fast) instead of the literal bisection (d_find).
enum DbgRawOp : int32_t { DBG_FIND = 0, DBG_FIND_FAST = 1, DBG_INSERT = 2, DBG_INSERT_FAST = 3, DBG_DELETE = 4, DBG_DELETE_FAST = 5,
                          DBG_PURGE = 6, DBG_REBALANCE = 7 };
hipError_t launch_dbg_raw_block(KeyArr keys, double* vals, uint64_t* occ, int64_t* sems, int64_t len, int op, int64_t key, double val,
                                int64_t from, int64_t to, int64_t m, int64_t* out, hipStream_t stream);
hipError_t launch_dbg_raw_wave(KeyArr keys, double* vals, uint64_t* occ, int64_t* sems, int64_t len, int op, int64_t key, double val,
                               int64_t from, int64_t to, int64_t m, int64_t* out, hipStream_t stream);
// device-side invariant checker; report[0..5] as documented at k_check_slots (8 x uint64 device scratch)
hipError_t launch_check(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, int64_t occ_words,
                        const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live, int64_t table_len,
                        unsigned long long* report, hipStream_t stream);
// view of one partition in one launch (ranges up to 16384 slots): meta = {from, to, err, partition id, cells or -1 = use the general path}
hipError_t launch_view_small(KeyArr keys, const double* vals, const uint64_t* occ, const int64_t* sems, const int64_t* col_keys,
                             const uint8_t* col_live, int64_t table_len, int64_t capacity, int64_t col, KeyArr out_k, double* out_v,
                             int64_t out_cap, int64_t* meta, int64_t* host, int64_t host_cells, unsigned long long seq, int64_t range_from,
                             int64_t range_to, hipStream_t stream);
// partition slot range lookup for views: out[0] = from (first slot after the semaphore), out[1] = to, or 0,0 if missing
hipError_t launch_partition_range(const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live,
                                  int64_t table_len, int64_t capacity, int64_t col, int64_t* out, hipStream_t stream);

// y = P x over one orientation P, gather form: y[part_key[p]] = sum over partition p of val * x[key]
hipError_t launch_spmv_gather(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                              const int64_t* sems, const int64_t* part_keys, const uint8_t* part_live, int64_t table_len,
                              const double* x, int64_t nx, double* y, int64_t ny, int pattern, int mode, hipStream_t stream);
constexpr int SPMV_ZFILL = 1;          // launch_spmv_gather mode bits (spmv.hip)
constexpr int SPMV_PLAIN_STREAM = 2;
constexpr int64_t SPMV_SPAN_SLOTS = 512;      // slots one wave of k_spmv_gather owns
constexpr int SPMV_META_BLOCKS = 1024;        // workgroups of k_spmv_meta at most (four table entries per thread and step)
constexpr int SPMV_META_WORDS = 3 * SPMV_META_BLOCKS + 1;     // its device scratch (zeroed once): partials, ticket
// out6_pinned: six words of PINNED host memory — the five results, then `seq` (written last, system scope: the host polls for it)
hipError_t launch_spmv_meta(const int64_t* sems, const int64_t* part_keys, int64_t table_len, int64_t capacity,
                            unsigned long long* scratch, unsigned long long* out6_pinned, unsigned long long seq, hipStream_t stream);
// y[key] += x[part_key[p]] * val, scatter form with fp64 atomics (the literal _mul loop nest)
hipError_t launch_spmv_scatter(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                               const int64_t* sems, const int64_t* part_keys, const uint8_t* part_live, int64_t table_len,
                               const double* x, int64_t nx, double* y, int64_t ny, hipStream_t stream);

// sparse x driven by its stored entries over the orientation whose partitions are x's index space (colmajor for mat*v):
// y (dense, zeroed here) += x_j * column j ; touched[row] = 1 for every row that received a term

}  // namespace dsa
