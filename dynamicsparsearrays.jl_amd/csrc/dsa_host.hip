// csrc/dsa_host.hip — host side of libdsa_hip.so: handles, HBM management, bulk construction
// (K-build), the yield loop around the on-device write sequencer, and the C ABI of include/dsa.h.
//
// Everything that touches slots runs on the GPU (rebalance.hip, sequencer.hip, spmv.hip).  The host
// keeps only: the control scalars of each PMA (mirrored from the device control block), the integer
// density bounds derived from the reference's Float64 thresholds (src/pma.jl:58,70,87 and :120-121),
// and the fill-mode staging buffer (src/buffer.jl — a host Dict in the reference as well).  The bulk builder
// (sort by (col,row), combine, emit, spread) runs on the device (build.hip + rebalance.hip).
// There is no CPU fallback for any slot operation.
#include "../../include/dsa.h"
#include "dsa_dev.h"

#include <dlfcn.h>
#include <sys/mman.h>
#ifndef MADV_HUGEPAGE
#define MADV_HUGEPAGE 14      /* <linux/mman.h>; hidden by the feature-test macros of this compilation */
#endif

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstddef>
#include <cstring>
#include <numeric>
#include <functional>
#include <mutex>
#include <exception>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

using namespace dsa;

namespace {

thread_local std::string g_err;

struct Fail { int32_t code; std::string msg; };
[[noreturn]] void fail(int32_t code, const std::string& msg) { throw Fail{code, msg}; }

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) fail(DSA_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

// roctx ranges around every ABI entry point (tracing): DSA_ROCTX=1 binds librocprofiler-sdk-roctx.so (or the legacy
// libroctx64.so) at run time, and `rocprofv3 --marker-trace` then shows which call a kernel belongs to; without the knob the
// cost is one predictable branch per call and no profiler library is mapped.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("DSA_ROCTX");
        if (!(e && e[0] == '1')) return;
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!lib) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
struct ApiRange {
    bool on;
    explicit ApiRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~ApiRange() { if (on) roctx().pop(); }
};

#define API_TRY ApiRange _api_range(__func__); try {
#define API_CATCH                                                              \
    } catch (const Fail& f) { g_err = f.msg; return f.code;                    \
    } catch (const std::bad_alloc&) { g_err = "host allocation failed"; return DSA_EHIP; \
    } catch (const std::exception& e) { g_err = e.what(); return DSA_EASSERT; } \
    return DSA_OK;

int g_device = 0;
std::atomic<bool> g_models_off{false};      // an append-replay model kernel could not be launched on this device (LDS): per-op replay only from then on
}  // namespace

// ---- the one table of development switches (dsa_dev.h: dev_env) --------------------------------------------------------------------
namespace dsa {
static const char* const k_dev_switches[] = {
    "DSA_APPEND_RUNS",    // 0: no append runs (per-op sequencer)
    "DSA_BARRIER_CHUNK",  // n: first sequencer chunk after a stop at an op that cannot be planned
    "DSA_BUILD_IDXSORT",  // 0: K-build always carries the values through the sort
    "DSA_BUILD_MINMAX",   // 1: key ranges by the device scan even when the host knows them
    "DSA_BUILD_TWIN",     // 0: the two orientations of a matrix are built independently from the triples
    "DSA_BUILD_WIDE",     // 1: K-build through the general (> 64-bit composite) path
    "DSA_BURST_GRAPH",    // 0: rounds as eager launches instead of a cached graph
    "DSA_COUNT_MODEL",    // 0: bitmap-only append replay (no model v2)
    "DSA_DBG_BURST", "DSA_DBG_RUN", "DSA_DBG_SPLIT", "DSA_DBG_SPMV", "DSA_DBG_SPMV_META", "DSA_DBG_TIME",      // timing / trace prints
    "DSA_DBG_MOVE2",      // ablations of the rebalance kernel (wrong results, timed)
    "DSA_FAIL_BUILD",     // 1: fails the next bulk build (fault injection, tests)
    "DSA_FP_MODE",        // footprint-check build (-DDSA_FP_CHECK): 1 recorded read / write sets, 2 sequential shadow re-plan (parbatch.hip)
    "DSA_KEYS_WIDE",      // 1: 64-bit physical keys everywhere
    "DSA_LOCAL_ROUNDS",   // 0: no local rounds (grid rounds only)
    "DSA_META_BLOCKS",    // workgroups of k_spmv_meta
    "DSA_MODEL3",         // 0: no count-only append replay
    "DSA_MODEL5",         // 0: no typed multi-level replay of 8-slot-segment append runs
    "DSA_MOVE2_BLOCK", "DSA_MOVE2_TILE",      // workgroup / tile size of k_move2
    "DSA_PARBATCH",       // 0: no batch-parallel rounds
    "DSA_POS_WIDE",       // 1: 64-bit positions in the append replay
    "DSA_PUBLISH",        // 0: device-to-host copies + stream synchronisation instead of the pinned hand-overs
    "DSA_RUN_AHEAD",      // 0: the rounds apply the conflict-free PREFIX only (rounds 2-5)
    "DSA_SEQ_CHUNK",      // n: first sequencer chunk after a stop by short prefixes
    "DSA_SMALL_BUILD",    // 0: small vectors through the general builder
    "DSA_SMALL_ROUNDS",   // 0: small matrix batches on two sequencers
    "DSA_SPMV_SHARE", "DSA_SPMV_STREAM", "DSA_SPMV_ZFILL", "DSA_SPMV_COMPACT",     // variants of the gather kernel
    "DSA_SPX_XDRIVEN",    // 0 / 1: the sparse-x product always through the gather kernel / always driven by x's entries
    "DSA_TIGHT",          // 0..3: tight footprints of leaf-accepted ops
    "DSA_TOMBSTONE_PAR",  // 0: orientations one after the other whenever tombstones exist
    "DSA_TWIN_ROUNDS",    // 0: the twin's deletes of deletecolumn! on a second sequencer
};
const char* dev_env(const char* name) {
#ifndef NDEBUG
    bool known = false;
    for (const char* s : k_dev_switches) known = known || std::strcmp(s, name) == 0;
    if (!known) { fprintf(stderr, "dev_env: %s is not in the table of development switches\n", name); abort(); }
#endif
#ifndef DSA_DEV
    const char* on = getenv("DSA_DEV");
    if (!(on && on[0] == '1')) return nullptr;
#endif
    return getenv(name);
}
}  // namespace dsa

namespace {
// dev knob: DSA_APPEND_RUNS=0 sends ascending append runs through the per-op sequencer path (A/B measurements)
// default of Pma::wait_policy (DSA_WAIT_POLICY=1: yield-friendly waits for every new handle)
const int g_wait_policy_default = [] { const char* e = getenv("DSA_WAIT_POLICY"); return (e && e[0] == '1') ? 1 : 0; }();
const bool g_append_runs = [] { const char* e = dev_env("DSA_APPEND_RUNS"); return !(e && e[0] == '0'); }();

// capacity = 2^ceil(Int, log2(ceil(n / t_h)))   src/pma.jl:64,81,88 (Float64 arithmetic, App. A.1)
int64_t capacity_for(int64_t n) {
    const double c = std::ceil((double)n / 0.7);
    const int64_t e = (int64_t)std::ceil(std::log2(c));
    return (int64_t)1 << e;
}

// ------------------------------------------------------------------------------------------------
// One packed-memory array resident in HBM, optionally with PackedCSC / MappedPackedCSC tables
// ------------------------------------------------------------------------------------------------
struct Pma {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* keys[2] = {nullptr, nullptr};      // physical key arrays: int32_t unless `wide` (KeyArr, dsa_dev.h)
    bool wide = false;
    double* vals[2] = {nullptr, nullptr};
    uint64_t* occ[2] = {nullptr, nullptr};
    int cur = 0;
    int64_t stat_why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t cap_alloc = 0;        // slots allocated per buffer
    int64_t occ_words = 0;        // words allocated per bitmap (whole 64-word tiles)
    int64_t occ_dirty[2] = {0, 0}; // high-water mark: words >= occ_dirty[b] of bitmap b are known to be zero
    bool has_sems = false, has_cols = false;
    int64_t* sems = nullptr; int64_t* col_keys = nullptr; uint8_t* col_live = nullptr;
    Ctl* d_ctl = nullptr;
    Ctl* h_ctl = nullptr;         // pinned host mirror
    RebalanceWork work{nullptr, nullptr, 0};
    RebalanceWork work2{nullptr, nullptr, 0};   // second prefix table of K-permute (old and new bitmap)
    uint64_t* occ_old = nullptr;                // bitmap saved by the sequencer at the start of an append run
    Op* run_cells = nullptr; uint64_t* run_flags = nullptr; int64_t* run_out = nullptr; int64_t run_cap = 0;   // cell stream of a MappedPackedCSC append run
    uint64_t* run_memo = nullptr;               // the append replay's memo between runs (sequencer.hip: k_append_run)
    Op* d_ops = nullptr; int64_t ops_cap = 0;
    int wait_policy = g_wait_policy_default;      // how blocking calls wait for a hand-over: 0 spin on the pinned word, 1 block in hipStreamSynchronize first (dsa_*_set_wait_policy)
    uint64_t* d_breaks = nullptr; bool breaks_valid = false;      // run-break bitmap of the ops in d_ops (sequencer.hip: k_op_breaks)
    int64_t* d_opsrc = nullptr; int64_t opsrc_cap = 0;            // the caller's columns of a batch (a, b, v: 3 x opsrc_cap x 8 B) before k_make_ops expands them
    double* d_q = nullptr; int64_t q_cap = 0;      // scratch for lookups (3 arrays of q_cap)
    int32_t* d_err = nullptr;
    int64_t stat_par_rounds = 0, stat_par_ops = 0, stat_seq_ops = 0, stat_seq_launches = 0;      // batch-parallel instrumentation
    int64_t stat_deferred = 0;              // ops a run-ahead round deferred behind a conflict (each is planned again in a later round)
    BurstGraph burst, burst_short;      // cached graphs of a full burst of rounds and of a short one (conflict-heavy phases)
    Plan* d_plans = nullptr; RoundState* d_rs = nullptr; RoundState* h_rs = nullptr;   // batch-parallel writes
    PendOp* d_pend = nullptr;                   // the pending lists of the rounds (2 x ROUND_GMAX: ops deferred behind a conflict, parbatch.hip)
    unsigned long long* h_pub = nullptr; unsigned int pub_seq = 0;      // pinned word k_publish writes the burst number to, and the last number handed out
    DevBufs* d_bufs = nullptr; DevBufs* h_bufs = nullptr;      // the arrays the rounds work on, read from device memory (pinned mirror)
    TableMerge tmerge{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; int64_t tmerge_cap = 0;   // scratch of the grid-wide table merge (tables.hip)
    int64_t stat_table_merges = 0;
    int64_t stat_grid_rebalances = 0;               // launches of the grid-wide rebalance (window_rebalance / root_rebalance)
    int64_t* d_small = nullptr;                     // 8 x int64 scratch
    int64_t* h_small = nullptr;                     // its pinned host mirror (small read-backs without a pageable staging copy)
    int64_t* h_get = nullptr; unsigned long long get_seq = 0;        // pinned landing area of small lookups (get_batch: keys, partitions, answers, error, sequence number)
    int64_t* h_view = nullptr; unsigned long long view_seq = 0;      // pinned landing area of column views: meta words, sequence number, first cells (col_view_of)
    hipEvent_t ev_handoff = nullptr;      // recorded on `stream` behind work another handle's stream must wait for (a slice built from this structure)
    // bumped by every launch that can move cells or change the tables; SpmvMeta is recomputed when it differs
    int device = 0;              // the device the handle lives on: re-selected at every API entry (a Julia task / finalizer thread or a
                                 // second Python thread calls in with whatever device its thread last selected)
    int64_t layout_epoch = 0;
    int64_t stat_spmv_nomemset = 0;
    struct SpmvMeta { int64_t epoch = -1; bool ordered = false; int64_t max_extent = 0, max_gap = 0, first_key = 0, last_key = 0; } spmv_meta;
    // its device side: scratch of k_spmv_meta, pinned landing area of the 5 result words, and the epoch a prefetch (enqueued behind
    // the write batch that changed the layout) is in flight for
    unsigned long long* d_meta = nullptr; int64_t* h_meta = nullptr; unsigned long long meta_seq = 0; int64_t meta_inflight_epoch = -1;
    // thresholds  src/pma.jl:58,70,87
    double t_h = 0.7, t_0 = 0.92, p_h = 0.3, p_0 = 0.08, t_d = 0.0, p_d = 0.0;

    int64_t capacity() const { return h_ctl->capacity; }
    KeyArr K() const { return KeyArr{keys[cur], wide ? 1 : 0, 0}; }
    KeyArr KA(int b) const { return KeyArr{keys[b], wide ? 1 : 0, 0}; }
    size_t kb() const { return wide ? sizeof(int64_t) : sizeof(int32_t); }
    double* V() const { return vals[cur]; }
    uint64_t* O() const { return occ[cur]; }
};

void pma_free_buffers(Pma& P) {
    for (int b = 0; b < 2; ++b) {
        pool_free(P.keys[b]); pool_free(P.vals[b]); pool_free(P.occ[b]);     // (the caller has synchronised the stream)
        P.keys[b] = nullptr; P.vals[b] = nullptr; P.occ[b] = nullptr;
    }
    pool_free(P.work.tile_cnt); pool_free(P.work.tile_off); pool_free(P.work.status);
    P.work = RebalanceWork{nullptr, nullptr, 0};
    pool_free(P.work2.tile_cnt); pool_free(P.work2.tile_off);
    P.work2 = RebalanceWork{nullptr, nullptr, 0};
    pool_free(P.occ_old);
    P.occ_old = nullptr;
}

void pma_destroy(Pma& P) {
    if (P.stream) hipStreamSynchronize(P.stream);
    pma_free_buffers(P);
    pool_free(P.sems); pool_free(P.col_keys); pool_free(P.col_live);
    pool_free(P.d_ctl);
    pinned_free(P.h_ctl);
    pool_free(P.d_ops); pool_free(P.d_breaks); pool_free(P.d_opsrc);      // (from the caching allocator since round 5: counted in DSA_INFO_HBM_BYTES)
    if (P.d_q) hipFree(P.d_q);
    pool_free(P.d_err);
    burst_graph_destroy(&P.burst);
    burst_graph_destroy(&P.burst_short);
    if (P.d_plans) hipFree(P.d_plans);
    if (P.d_pend) hipFree(P.d_pend);
    if (P.d_bufs) hipFree(P.d_bufs);
    if (P.h_bufs) hipHostFree(P.h_bufs);
    if (P.d_rs) hipFree(P.d_rs);
    if (P.h_rs) hipHostFree(P.h_rs);
    pool_free(P.d_small);
    pinned_free(P.h_small);
    pinned_free(P.h_view);
    if (P.ev_handoff) (void)hipEventDestroy(P.ev_handoff);
    if (P.d_meta) hipFree(P.d_meta);
    pinned_free(P.h_meta);
    if (P.tmerge.sems2) hipFree(P.tmerge.sems2);
    if (P.tmerge.keys2) hipFree(P.tmerge.keys2);
    if (P.tmerge.pkey) hipFree(P.tmerge.pkey);
    if (P.run_cells) hipFree(P.run_cells);
    if (P.run_flags) hipFree(P.run_flags);
    if (P.run_out) hipFree(P.run_out);
    if (P.run_memo) hipFree(P.run_memo);
    if (P.own_stream && P.stream) stream_put(P.stream, P.device);      // synchronised at the top of this function
    P = Pma();
}

// keys cross the host boundary as int64_t; the device array is int32_t unless the structure is wide
void upload_keys(Pma& P, void* dst, const int64_t* src, int64_t n) {
    if (n <= 0) return;
    if (P.wide) { HIPCHK(hipMemcpyAsync(dst, src, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, P.stream)); HIPCHK(hipStreamSynchronize(P.stream)); return; }
    std::vector<int32_t> tmp((size_t)n);
    for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = (int32_t)src[i];
    HIPCHK(hipMemcpyAsync(dst, tmp.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
}
void download_keys(Pma& P, int64_t* dst, const void* src, int64_t n) {      // synchronises the stream
    if (n <= 0) { HIPCHK(hipStreamSynchronize(P.stream)); return; }
    if (P.wide) { HIPCHK(hipMemcpyAsync(dst, src, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, P.stream)); HIPCHK(hipStreamSynchronize(P.stream)); return; }
    std::vector<int32_t> tmp((size_t)n);
    HIPCHK(hipMemcpyAsync(tmp.data(), src, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    for (int64_t i = 0; i < n; ++i) dst[i] = (int64_t)tmp[(size_t)i];
}

// dev knob: DSA_KEYS_WIDE=1 keeps every structure in 64-bit keys (A/B measurements, coverage of the wide kernels)
const bool g_force_wide = [] { const char* e = dev_env("DSA_KEYS_WIDE"); return e && e[0] == '1'; }();
bool keys_fit32(const int64_t* k, int64_t n) {
    if (g_force_wide) return false;
    for (int64_t i = 0; i < n; ++i) if (!key_fits32(k[i])) return false;
    return true;
}

// one pass over a key array the host is about to upload: value range (what K-build's composite needs), storage width, the reserved key
struct KeyScan {
    int64_t lo = INT64_MAX, hi = INT64_MIN; bool zero = false;
    void add(int64_t k) { lo = k < lo ? k : lo; hi = k > hi ? k : hi; zero = zero || k == 0; }
    void add(const int64_t* k, int64_t n) {
        int64_t l = lo, h = hi; bool z = zero;
        for (int64_t i = 0; i < n; ++i) { const int64_t v = k[i]; l = v < l ? v : l; h = v > h ? v : h; z |= v == 0; }
        lo = l; hi = h; zero = z;
    }
    bool empty() const { return hi < lo; }
    bool fit32() const { return !g_force_wide && (empty() || (key_fits32(lo) && key_fits32(hi))); }
    KeyRange range() const { KeyRange r; if (!empty()) { r.lo = lo; r.hi = hi; } return r; }
};

int64_t occ_words_for(int64_t slots) {
    const int64_t w = (slots + 63) / 64;
    return ((w + 63) / 64) * 64;      // whole 64-word tiles (k_tile_count / k_move read lane <-> word)
}

// slot buffers come from the caching allocator (pool.hip): a structure built after another one of the same size was destroyed
// finds its ~100 MB blocks again without a driver call
void alloc_one_buffer(Pma& P, int b, int64_t slots, bool zero = true) {
    HIPCHK(pool_alloc(&P.keys[b], (size_t)slots * P.kb()));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.vals[b]), (size_t)slots * sizeof(double)));
    const int64_t words = occ_words_for(slots);
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.occ[b]), (size_t)words * sizeof(uint64_t)));
    if (zero) HIPCHK(hipMemsetAsync(P.occ[b], 0, (size_t)words * sizeof(uint64_t), P.stream));
}

// zero = false: the caller zeroes the bitmaps and the status table itself (launch_init_fresh: one launch for all of them)
void alloc_work(Pma& P, int64_t slots, bool zero = true) {
    // (from the caching allocator since round 6: five driver allocations per new structure were a third of what a small vector — a
    //  slice, a filter result — costs to create; the caller has waited for the stream before an existing table is replaced)
    pool_free(P.work.tile_cnt); pool_free(P.work.tile_off); pool_free(P.work.status);
    P.work = RebalanceWork{nullptr, nullptr, 0};
    P.work.tiles_cap = slots / 4096 + 8;
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.work.tile_cnt), (size_t)P.work.tiles_cap * sizeof(uint32_t)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.work.tile_off), (size_t)P.work.tiles_cap * sizeof(uint32_t)));
    P.work.status_cap = slots / 1024 + slots / (1024 * 64) + 16; P.work.gen = 0;      // one word per 1024-slot tile + one per 64 tiles + the fault word
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.work.status), (size_t)P.work.status_cap * sizeof(unsigned long long)));
    if (zero) HIPCHK(hipMemsetAsync(P.work.status, 0, (size_t)P.work.status_cap * sizeof(unsigned long long), P.stream));
    pool_free(P.work2.tile_cnt); pool_free(P.work2.tile_off);
    P.work2 = RebalanceWork{nullptr, nullptr, 0};
    P.work2.tiles_cap = P.work.tiles_cap;
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.work2.tile_cnt), (size_t)P.work2.tiles_cap * sizeof(uint32_t)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.work2.tile_off), (size_t)P.work2.tiles_cap * sizeof(uint32_t)));
    pool_free(P.occ_old); P.occ_old = nullptr;
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.occ_old), (size_t)occ_words_for(slots) * sizeof(uint64_t)));
}

void bind_device(const Pma& P) { HIPCHK(hipSetDevice(P.device)); }

void pma_init_common(Pma& P, bool sems, bool cols) {
    HIPCHK(hipSetDevice(g_device));
    P.device = g_device;
    // stream, control blocks and landing areas come from the caches of pool.hip: a handle is created without a driver call once
    // another one has died (0.5 ms per PMA otherwise: two per matrix, inside every closefillmode! / dynamicsparse)
    HIPCHK(stream_get(&P.stream));
    P.own_stream = true;
    P.has_sems = sems; P.has_cols = cols;
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.d_ctl), sizeof(Ctl)));
    HIPCHK(pinned_alloc(reinterpret_cast<void**>(&P.h_ctl), sizeof(Ctl)));
    std::memset(P.h_ctl, 0, sizeof(Ctl));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.d_err), sizeof(int32_t)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.d_small), 8 * sizeof(int64_t)));
    // one pinned block of 4 KB (the allocator's smallest class) per structure: [0, 64) small read-backs, [64, 72) the word the publish
    // kernels write their number to, [128, 128 + 2 KB) the landing area of small lookups — a structure costs no further pinned blocks
    // (a program with 10^5 small vectors pays 4 KB of pinned memory for each, not 12)
    HIPCHK(pinned_alloc(reinterpret_cast<void**>(&P.h_small), 4096));
    std::memset(P.h_small, 0, 4096);
    P.h_pub = reinterpret_cast<unsigned long long*>(P.h_small + 8);
    P.h_get = P.h_small + 16;
}

void ensure_tables(Pma& P, int64_t need) {
    if (!P.has_sems) return;
    if (need <= P.h_ctl->table_cap) return;
    int64_t ncap = std::max<int64_t>(1024, P.h_ctl->table_cap * (P.h_ctl->table_cap < (1 << 20) ? 4 : 2));      // 4x steps below 1 M entries, 2x above
    while (ncap < need) ncap *= ncap < (1 << 20) ? 4 : 2;
    int64_t* ns = nullptr; int64_t* nk = nullptr; uint8_t* nl = nullptr;
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&ns), (size_t)ncap * sizeof(int64_t)));
    HIPCHK(hipMemsetAsync(ns, 0, (size_t)ncap * sizeof(int64_t), P.stream));
    const int64_t len = P.h_ctl->table_len;
    if (P.sems && len > 0) HIPCHK(hipMemcpyAsync(ns, P.sems, (size_t)len * sizeof(int64_t), hipMemcpyDeviceToDevice, P.stream));
    if (P.has_cols) {
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&nk), (size_t)ncap * sizeof(int64_t)));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&nl), (size_t)ncap));
        HIPCHK(hipMemsetAsync(nk, 0, (size_t)ncap * sizeof(int64_t), P.stream));
        HIPCHK(hipMemsetAsync(nl, 0, (size_t)ncap, P.stream));
        if (P.col_keys && len > 0) {
            HIPCHK(hipMemcpyAsync(nk, P.col_keys, (size_t)len * sizeof(int64_t), hipMemcpyDeviceToDevice, P.stream));
            HIPCHK(hipMemcpyAsync(nl, P.col_live, (size_t)len, hipMemcpyDeviceToDevice, P.stream));
        }
    }
    // (fresh tables: the memsets above are stream-ordered in front of whatever uses them — no wait; the K-build sizes its tables while
    //  its sort kernels run on this stream, and a wait here would be a wait for the sort)
    if (P.sems || P.col_keys || P.col_live) HIPCHK(hipStreamSynchronize(P.stream));
    pool_free(P.sems); pool_free(P.col_keys); pool_free(P.col_live);
    P.sems = ns; P.col_keys = nk; P.col_live = nl;
    P.h_ctl->table_cap = ncap;
}

// integer density bounds of every level (see Ctl) from the reference's Float64 thresholds
void compute_bounds(Pma& P) {
    Ctl& c = *P.h_ctl;
    if (c.height + 1 > MAX_LEVELS) fail(DSA_EARG, "PMA too tall");
    P.t_d = (P.t_h - P.t_0) / (double)c.height;      // src/pma.jl:47-48,147-148,157-158
    P.p_d = (P.p_h - P.p_0) / (double)c.height;
    for (int64_t h = 0; h <= c.height; ++h) {
        const double W = (double)(c.segment_capacity << h);
        volatile double pm = P.p_d * (double)h;        // separate multiply and add, as Julia evaluates them
        volatile double tm = P.t_d * (double)h;
        const double p = P.p_0 + pm;
        const double t = P.t_0 + tm;
        c.lo[h] = (int64_t)std::ceil(p * W);           // p <= count/W  <=>  count >= ceil(p*W)   (W = 2^k: exact)
        c.hi[h] = (int64_t)std::floor(t * W);          // count/W <= t  <=>  count <= floor(t*W)
    }
}

// _pma geometry  src/pma.jl:42-49
void set_geometry_for_new(Pma& P, int64_t capacity, int64_t nb_elements) {
    Ctl& c = *P.h_ctl;
    const double lc = std::log2((double)capacity);
    const int64_t nb_segs = (int64_t)1 << (int64_t)std::ceil(std::log2((double)capacity / lc));
    c.capacity = capacity;
    c.nb_segments = nb_segs;
    c.segment_capacity = capacity / nb_segs;
    c.height = (int64_t)std::log2((double)nb_segs);
    c.nb_elements = nb_elements;
    compute_bounds(P);
}

void upload_ctl(Pma& P) {
    HIPCHK(hipMemcpyAsync(P.d_ctl, P.h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));   // h_ctl is reused as the download target
}
void download_ctl(Pma& P) {
    HIPCHK(hipMemcpyAsync(P.h_ctl, P.d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
}

// grow both slot buffers to at least `slots` (contents of the current buffer are preserved)
void ensure_capacity_alloc(Pma& P, int64_t slots, bool zero = true) {
    if (slots <= P.cap_alloc) return;
    // growth in steps of 4x (at least 64k slots once the first 4096 are outgrown): a growing array re-allocates its two buffers
    // (13 hipMalloc / hipFree and a stream wait each time) 4 times on the way to 4M slots instead of 10; HBM is not the scarce resource
    // ... up to 2^24 slots; above that the steps are 2x (a structure one slot past a 4x boundary would otherwise hold 4x what it
    // needs twice over: 2^26 + 1 slots -> 2 x 2^28 x 12 B)
    int64_t n = std::max<int64_t>(P.cap_alloc, 4096);
    if (n < slots) n = std::max<int64_t>(n < (1 << 24) ? 4 * n : 2 * n, 65536);
    while (n < slots) n *= n < (1 << 24) ? 4 : 2;
    void* ok[2] = {P.keys[0], P.keys[1]}; double* ov[2] = {P.vals[0], P.vals[1]}; uint64_t* oo[2] = {P.occ[0], P.occ[1]};
    const int64_t old_words = P.occ_words, old_slots = P.cap_alloc;
    for (int b = 0; b < 2; ++b) { P.keys[b] = nullptr; P.vals[b] = nullptr; P.occ[b] = nullptr; }
    for (int b = 0; b < 2; ++b) alloc_one_buffer(P, b, n, zero);
    if (ok[P.cur] != nullptr && old_slots > 0) {
        HIPCHK(hipMemcpyAsync(P.keys[P.cur], ok[P.cur], (size_t)old_slots * P.kb(), hipMemcpyDeviceToDevice, P.stream));
        HIPCHK(hipMemcpyAsync(P.vals[P.cur], ov[P.cur], (size_t)old_slots * sizeof(double), hipMemcpyDeviceToDevice, P.stream));
        HIPCHK(hipMemcpyAsync(P.occ[P.cur], oo[P.cur], (size_t)old_words * sizeof(uint64_t), hipMemcpyDeviceToDevice, P.stream));
    }
    if (ok[0] || ok[1]) HIPCHK(hipStreamSynchronize(P.stream));      // (old buffers: copied out of and about to be freed; a fresh array waits for nobody)
    for (int b = 0; b < 2; ++b) { pool_free(ok[b]); pool_free(ov[b]); pool_free(oo[b]); }
    P.occ_dirty[1 - P.cur] = 0;                       // fresh, zero-filled; occ_dirty[cur] keeps its value
    P.cap_alloc = n;
    P.occ_words = occ_words_for(n);
    alloc_work(P, n, zero);
}

// pack + spread of the whole array into the other buffer: cells of cur[1..src_cap] -> alt[1..new_cap]
// (root _even_rebalance!, _extend!, pack! + _shrink!)  src/pma.jl:94-103,135-161
void root_rebalance(Pma& P, int64_t src_cap, int64_t new_cap, int64_t m, bool src_packed) {
    ++P.stat_grid_rebalances;
    ensure_capacity_alloc(P, std::max(src_cap, new_cap));
    ++P.layout_epoch;
    const int alt = 1 - P.cur;
    hipError_t e = launch_rebalance(P.KA(P.cur), P.vals[P.cur], P.occ[P.cur], 1, src_cap, src_packed,
                                    P.KA(alt), P.vals[alt], P.occ[alt], 1, new_cap, m,
                                    P.has_sems ? P.sems : nullptr, &P.work, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("rebalance launch: ") + hipGetErrorString(e));
    // bits beyond the new capacity must be zero in the buffer that becomes current; only the words that
    // may still hold stale bits (below the buffer's high-water mark) are cleared
    const int64_t first_word = (new_cap + 63) / 64;
    if (first_word < P.occ_dirty[alt])
        HIPCHK(hipMemsetAsync(P.occ[alt] + first_word, 0, (size_t)(P.occ_dirty[alt] - first_word) * sizeof(uint64_t), P.stream));
    P.occ_dirty[alt] = first_word;
    P.cur = alt;
}

// an interior window (too wide for the LDS paths): pack! into the alternate buffer, spread! back from there — the two halves of
// _even_rebalance! (src/pma.jl:94-103) as two launches of the same kernel: unpacked source -> m packed cells, packed source ->
// spread window.  (2 W + 2 m) cells of traffic and two launches; round 2 rebalanced into the alternate buffer and copied the
// window back with three device-to-device copies: 4 W cells, four launches.)
void window_rebalance(Pma& P, int64_t ws, int64_t we, int64_t m) {
    if (ws == 1 && we == P.capacity()) { root_rebalance(P, P.capacity(), P.capacity(), m, false); return; }
    const int alt = 1 - P.cur;
    ++P.layout_epoch;
    ++P.stat_grid_rebalances;
    if (m <= 0) {                                     // nothing to move: every slot of the window becomes a gap
        hipError_t e0 = launch_clear_occ(P.O(), ws, we, P.stream);
        if (e0 != hipSuccess) fail(DSA_EHIP, std::string("clear launch: ") + hipGetErrorString(e0));
        return;
    }
    // pack!: the m cells of [ws, we] -> alt[ws .. ws + m - 1] (no gaps: the destination window has exactly m slots); the semaphore
    // table is not touched (positions in the scratch buffer mean nothing)
    hipError_t e = launch_rebalance(P.K(), P.V(), P.O(), ws, we, false, P.KA(alt), P.vals[alt], P.occ[alt], ws, ws + m - 1, m,
                                    nullptr, &P.work, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("rebalance launch (pack): ") + hipGetErrorString(e));
    P.occ_dirty[alt] = std::max<int64_t>(P.occ_dirty[alt], (ws + m - 1 + 63) / 64);      // the scratch bitmap words written by the pack
    // spread!: packed source -> the window in the current buffer, occupancy words and semaphores[] included
    e = launch_rebalance(P.KA(alt), P.vals[alt], P.occ[alt], ws, ws + m - 1, true, P.K(), P.V(), P.O(), ws, we, m,
                         P.has_sems ? P.sems : nullptr, &P.work, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("rebalance launch (spread): ") + hipGetErrorString(e));
}

// PackedMemoryArray(keys, values; sort=false) + _pma  src/pma.jl:42-55,69-84 from an already ordered
// cell stream; n == 0 -> PackedMemoryArray(K, T) (capacity for 100 expected cells)  src/pma.jl:86-91
void build_from_packed(Pma& P, const std::vector<int64_t>& keys, const std::vector<double>& vals) {
    const int64_t n = (int64_t)keys.size();
    if (P.cap_alloc == 0) P.wide = !keys_fit32(keys.data(), n);
    const int64_t capacity = capacity_for(n == 0 ? 100 : n);
    set_geometry_for_new(P, capacity, n);
    ensure_capacity_alloc(P, 2 * capacity);
    if (n > 0) {
        upload_keys(P, P.keys[P.cur], keys.data(), n);
        HIPCHK(hipMemcpyAsync(P.V(), vals.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, P.stream));
    }
    // _even_rebalance!(pma, 1, capacity, n): a no-op when the array is exactly one leaf (src/pma.jl:96-99)
    P.h_ctl->stat_rebalances = 0; P.h_ctl->stat_window_slots = 0;
    if (capacity != P.h_ctl->segment_capacity) { P.h_ctl->stat_rebalances = 1; P.h_ctl->stat_window_slots = capacity; }
    root_rebalance(P, std::max<int64_t>(n, 1), capacity, n, true);
    upload_ctl(P);
}

// An append run was simulated on the bitmap of the current buffer (sequencer.hip): the y_we cells that existed before the
// run (positions: saved bitmap occ_old) followed by the cells cells[i0..] move to the set bits of the current bitmap,
// written into the alternate buffer, which becomes current.
void permute_run(Pma& P, const Op* cells, int64_t i0, int64_t n0) {
    const int alt = 1 - P.cur;
    const int64_t cap = P.capacity();
    ++P.layout_epoch;
    hipError_t e = launch_permute(P.K(), P.V(), P.occ_old, cap, P.KA(alt), P.vals[alt], P.O(), cap, n0, cells, i0,
                                  P.has_sems ? P.sems : nullptr, &P.work, &P.work2, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("permute launch: ") + hipGetErrorString(e));
    const int64_t words = (cap + 63) / 64;
    HIPCHK(hipMemcpyAsync(P.occ[alt], P.O(), (size_t)words * sizeof(uint64_t), hipMemcpyDeviceToDevice, P.stream));
    if (words < P.occ_dirty[alt])
        HIPCHK(hipMemsetAsync(P.occ[alt] + words, 0, (size_t)(P.occ_dirty[alt] - words) * sizeof(uint64_t), P.stream));
    P.occ_dirty[alt] = words;
    P.cur = alt;
}

// A batch of ops as the host hands it to a structure: a ready-made Op array (small batches, mixed kinds), or the caller's COLUMNS —
// op k = (a[k], b ? b[k] : 0, v[k]) of one kind — which go up as they are (16 / 24 bytes per op instead of 32, no Op vector built on
// the host) and are expanded into the op array by a kernel behind the upload (sequencer.hip: k_make_ops).
struct OpBatch {
    int64_t n = 0;
    const Op* ops = nullptr;
    const int64_t* a = nullptr; const int64_t* b = nullptr; const double* v = nullptr; int32_t kind = 0;
    OpBatch() = default;
    OpBatch(const std::vector<Op>& o) : n((int64_t)o.size()), ops(o.data()) {}      // NOLINT: implicit by design
    OpBatch(int32_t kind_, const int64_t* a_, const int64_t* b_, const double* v_, int64_t n_) : n(n_), a(a_), b(b_), v(v_), kind(kind_) {}
    Op at(int64_t k) const {
        if (ops) return ops[k];
        Op o; o.a = a[k]; o.b = b ? b[k] : 0; o.v = v[k]; o.kind = kind; o.pad = 0; return o;
    }
    int64_t key(int64_t k) const { return ops ? ops[k].a : a[k]; }
};

void ensure_ops(Pma& P, int64_t n) {
    P.breaks_valid = false;
    if (n <= P.ops_cap) return;
    HIPCHK(hipStreamSynchronize(P.stream));                  // (a pooled block is handed out again at once: nothing may still read the old one)
    pool_free(P.d_ops); pool_free(P.d_breaks);
    P.d_ops = nullptr; P.d_breaks = nullptr;
    P.ops_cap = std::max<int64_t>(n, 1024);
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.d_ops), (size_t)P.ops_cap * sizeof(Op)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.d_breaks), (size_t)(P.ops_cap / 64 + 8) * sizeof(uint64_t)));
}
// the ops of a batch into d_ops (stream-ordered; the host arrays must stay alive until the batch has finished — every batch waits)
void upload_batch(Pma& P, const OpBatch& B) {
    const int64_t n = B.n;
    if (B.ops != nullptr) {
        HIPCHK(hipMemcpyAsync(P.d_ops, B.ops, (size_t)n * sizeof(Op), hipMemcpyHostToDevice, P.stream));
        return;
    }
    if (n > P.opsrc_cap) {
        HIPCHK(hipStreamSynchronize(P.stream));
        pool_free(P.d_opsrc); P.d_opsrc = nullptr;
        P.opsrc_cap = std::max<int64_t>(n, P.ops_cap);
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&P.d_opsrc), (size_t)P.opsrc_cap * 3 * sizeof(int64_t)));
    }
    int64_t* da = P.d_opsrc; int64_t* db = P.d_opsrc + P.opsrc_cap; double* dv = reinterpret_cast<double*>(P.d_opsrc + 2 * P.opsrc_cap);
    HIPCHK(hipMemcpyAsync(da, B.a, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, P.stream));
    if (B.b) HIPCHK(hipMemcpyAsync(db, B.b, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, P.stream));
    HIPCHK(hipMemcpyAsync(dv, B.v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, P.stream));
    hipError_t e = launch_make_ops(da, B.b ? db : nullptr, dv, B.kind, n, P.d_ops, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("make ops launch: ") + hipGetErrorString(e));
}

// the n ops just uploaded into d_ops: where an append run cannot continue (read by the sequencer's run detection), enqueued behind the
// upload.  Vectors and MappedPackedCSC only — a plain PackedCSC has no runs
void enqueue_op_breaks(Pma& P, int64_t n) {
    P.breaks_valid = false;
    if (!g_append_runs || P.occ_old == nullptr || n < 64 || (P.has_sems && !P.has_cols)) return;
    hipError_t e = launch_op_breaks(P.d_ops, n, P.has_cols ? 1 : 0, P.d_breaks, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("op breaks launch: ") + hipGetErrorString(e));
    P.breaks_valid = true;
}

int32_t seq_err_to_status(int32_t e) { return e == 0 ? DSA_EASSERT : e; }

const char* err_text(int32_t e) {
    switch (e) {
        case DSA_EARG: return "column does not exist.";
        case DSA_EBOUNDS: return "cannot access partition at this index";
        case DSA_EDELETED: return "The partition has been deleted.";
        case DSA_EFULL: return "No empty cell to insert a new element.";
        case DSA_EASSERT: return "reference assertion failed (tombstoned partition in the way)";
        default: return "sequencer error";
    }
}

// First key outside Int32: both slot buffers are re-allocated with 64-bit keys, the current one converted on the device.
// (The alternate buffer holds no live data between operations.)
void widen_keys(Pma& P) {
    if (P.wide) return;
    HIPCHK(hipStreamSynchronize(P.stream));
    void* old[2] = {P.keys[0], P.keys[1]};
    for (int b = 0; b < 2; ++b) { P.keys[b] = nullptr; if (P.cap_alloc > 0) HIPCHK(pool_alloc(&P.keys[b], (size_t)P.cap_alloc * sizeof(int64_t))); }
    if (P.cap_alloc > 0 && old[P.cur] != nullptr) {
        hipError_t e = launch_widen_keys(old[P.cur], P.keys[P.cur], P.cap_alloc, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("widen launch: ") + hipGetErrorString(e));
        HIPCHK(hipStreamSynchronize(P.stream));
    }
    for (int b = 0; b < 2; ++b) pool_free(old[b]);
    P.wide = true;
}
void ensure_key_width(Pma& P, const OpBatch& B) {
    if (P.wide) return;
    if (B.ops != nullptr) {
        for (int64_t k = 0; k < B.n; ++k) {
            const Op& o = B.ops[k];
            if ((o.kind == OP_VEC_SET || o.kind == OP_PCSC_SET || o.kind == OP_MPCSC_SET) && !key_fits32(o.a)) { widen_keys(P); return; }
        }
        return;
    }
    for (int64_t k = 0; k < B.n; ++k) if (!key_fits32(B.a[k])) { widen_keys(P); return; }
}
void ensure_key_width(Pma& P, const std::vector<Op>& ops) {
    if (P.wide) return;
    for (const Op& o : ops)
        if ((o.kind == OP_VEC_SET || o.kind == OP_PCSC_SET || o.kind == OP_MPCSC_SET) && !key_fits32(o.a)) { widen_keys(P); return; }
}

// ---- the yield loop around the device sequencer, as a resumable state machine so that the two orientations of a
// matrix can run their sequencers concurrently on their own streams ------------------------------------------------
// Pending partition-table entries (created by the running batch at the end of the tables, Ctl::n_pending) back into key order:
// the grid-wide pass of tables.hip, stream-ordered, no host wait.  h_ctl->table_cap must be current.
void merge_tables(Pma& P) {
    if (!P.has_cols) return;
    const int64_t cap = P.h_ctl->table_cap;
    if (P.tmerge_cap < cap) {
        if (P.tmerge.sems2) HIPCHK(hipFree(P.tmerge.sems2));          // hipFree waits for the work that may still use them
        if (P.tmerge.keys2) HIPCHK(hipFree(P.tmerge.keys2));
        P.tmerge.sems2 = P.tmerge.keys2 = nullptr; P.tmerge_cap = 0;
        HIPCHK(hipMalloc(&P.tmerge.sems2, (size_t)cap * sizeof(int64_t)));
        HIPCHK(hipMalloc(&P.tmerge.keys2, (size_t)cap * sizeof(int64_t)));
        P.tmerge_cap = cap;
    }
    if (!P.tmerge.pkey) {
        HIPCHK(hipMalloc(&P.tmerge.pkey, (size_t)(3 * 1024 + 8) * sizeof(int64_t)));
        P.tmerge.pdst = P.tmerge.pkey + 1024; P.tmerge.psem = P.tmerge.pkey + 2048; P.tmerge.hdr = P.tmerge.pkey + 3072;
    }
    ++P.layout_epoch;
    if (P.h_ctl->n_pending > TABLE_PEND_MAX) fail(DSA_EASSERT, "more pending partition-table entries than the merge takes (internal invariant)");
    hipError_t e = launch_table_merge(P.sems, P.col_keys, P.col_live, P.V(), P.d_ctl, P.tmerge, cap, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("table merge launch: ") + hipGetErrorString(e));
    P.h_ctl->n_pending = 0;
    P.stat_table_merges += 1;
}

struct SeqRun {
    Pma* P = nullptr;
    const std::vector<Op>* ops = nullptr;
    int64_t n = 0;
    int64_t n_avail = 0;     // ops resident in d_ops (>= n): an append run may consume ops beyond the chunk
    bool active = false;
    int32_t err = 0;         // status of the failing op (0 if none)
    int64_t applied = 0;     // ops fully applied
    int64_t guard = 0;
    bool defer_merge = false; // leave pending table entries to the caller (a batch that goes on with more launches)
};

// Hand-over of a launch's result through pinned memory (parbatch.hip: k_publish; the sequencer does it in its own epilogue): the last kernel of the launch
// writes the control block (and the round state) into the host's pinned mirrors and then a number into P.h_pub; the host polls for
// that number instead of issuing device-to-host copies and synchronising the stream.  DSA_PUBLISH=0: copies + synchronisation.
bool publish_enabled() { static const bool on = [] { const char* e = dev_env("DSA_PUBLISH"); return !(e && e[0] == '0'); }(); return on; }
unsigned int next_publish_seq(Pma& P) {
    if (++P.pub_seq == 0) P.pub_seq = 1;
    return P.pub_seq;
}
// Blocking calls wait for a word the last kernel of the launch writes into pinned memory.  Policy 0 polls it (lowest latency; the
// calling thread spins on a host core for the microseconds to milliseconds the device needs).  Policy 1 parks the thread in
// hipStreamSynchronize first — the word is there when it returns — for hosts that run many tasks on few threads (a Julia process
// driving Coluna): the kernels, the hand-over and the results are the same, only the way the host waits differs.
void wait_policy_block(Pma& P) {
    if (P.wait_policy == 1) HIPCHK(hipStreamSynchronize(P.stream));
}
void wait_published(Pma& P) {
    wait_policy_block(P);
    // the stream is asked now and then so that a failed launch or a faulted kernel cannot hang the host
    volatile unsigned long long* seqp = P.h_pub;
    auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    while ((unsigned int)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != P.pub_seq) {
        if (std::chrono::steady_clock::now() < next_query) continue;
        const hipError_t q = hipStreamQuery(P.stream);
        if (q == hipErrorNotReady) { next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2); continue; }
        if (q != hipSuccess) fail(DSA_EHIP, std::string("device work failed: ") + hipGetErrorString(q));
        if ((unsigned int)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != P.pub_seq) fail(DSA_EHIP, "device work finished without publishing its state");
    }
}

static thread_local double g_seq_launch_ms = 0;
void seq_launch(SeqRun& r, bool upload = true) {
    Pma& P = *r.P;
    struct T { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
               ~T() { g_seq_launch_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } timer;
    // pinned h_ctl: H2D, kernel and D2H are stream-ordered; the host does not touch h_ctl until the next synchronize
    if (upload) HIPCHK(hipMemcpyAsync(P.d_ctl, P.h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, P.stream));
    ++P.layout_epoch;
    const bool publish = publish_enabled();
    const unsigned int seq = publish ? next_publish_seq(P) : 0u;       // the sequencer hands its control block back itself
    hipError_t e = launch_sequencer(P.K(), P.V(), P.O(), P.has_sems ? P.sems : nullptr, P.has_cols ? P.col_keys : nullptr,
                                    P.has_cols ? P.col_live : nullptr, P.d_ctl, P.d_ops, r.n, std::max(r.n, r.n_avail),
                                    g_append_runs && P.occ_old != nullptr, P.breaks_valid ? P.d_breaks : nullptr,
                                    publish ? P.h_ctl : nullptr, publish ? P.h_pub : nullptr, seq, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("sequencer launch: ") + hipGetErrorString(e));
    if (!publish) HIPCHK(hipMemcpyAsync(P.h_ctl, P.d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, P.stream));
}

void seq_start(SeqRun& r, Pma& P, const std::vector<Op>& ops) {
    r = SeqRun();
    r.P = &P; r.ops = &ops; r.n = (int64_t)ops.size();
    if (r.n == 0) return;
    ensure_key_width(P, ops);
    ensure_ops(P, r.n);
    HIPCHK(hipMemcpyAsync(P.d_ops, ops.data(), (size_t)r.n * sizeof(Op), hipMemcpyHostToDevice, P.stream));
    enqueue_op_breaks(P, r.n);
    P.h_ctl->next_op = 0; P.h_ctl->status = 0; P.h_ctl->err = 0; P.h_ctl->no_run_at = -1;
    r.active = true;
    seq_launch(r);
}

// waits for the running kernel of `r`, services its yield and relaunches; returns false once the batch is finished
// dev (DSA_DBG_SPLIT): where a sequencer chunk spends its wall clock — waiting for the device / host work per kind of yield
static thread_local double g_seq_wait_ms = 0, g_seq_host_ms[8] = {0};
struct SeqStepTimer {
    std::chrono::steady_clock::time_point t0; int kind;
    SeqStepTimer(int k) : t0(std::chrono::steady_clock::now()), kind(k) {}
    ~SeqStepTimer() { g_seq_host_ms[kind & 7] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
bool seq_step(SeqRun& r) {
    if (!r.active) return false;
    Pma& P = *r.P;
    {
        const auto tw0 = std::chrono::steady_clock::now();
        if (publish_enabled()) wait_published(P); else HIPCHK(hipStreamSynchronize(P.stream));
        g_seq_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
    }
    Ctl& c = *P.h_ctl;
    SeqStepTimer timer(c.status);
    switch (c.status) {
        case SEQ_DONE:
            r.applied = std::max(r.n, c.next_op); r.active = false;
            if (!r.defer_merge && c.n_pending > 0) merge_tables(P);
            return false;
        case SEQ_ERROR:
            r.err = seq_err_to_status(c.err); r.applied = c.next_op; r.active = false;
            if (!r.defer_merge && c.n_pending > 0) merge_tables(P);
            return false;
        case SEQ_Y_REBALANCE:
            window_rebalance(P, c.y_ws, c.y_we, c.y_m);
            break;
        case SEQ_Y_EXTEND: {       // _extend!  src/pma.jl:143-151 then _even_rebalance!(1, capacity, count)
            const int64_t old_cap = c.capacity;
            c.capacity *= 2; c.nb_segments *= 2; c.height += 1;
            compute_bounds(P);
            c.stat_extends += 1; c.stat_rebalances += 1; c.stat_window_slots += c.capacity;
            root_rebalance(P, old_cap, c.capacity, c.y_m, false);
            break;
        }
        case SEQ_Y_SHRINK: {       // pack! + _shrink!  src/pma.jl:135-139,153-161 then _even_rebalance!
            const int64_t old_cap = c.capacity;
            c.capacity /= 2; c.nb_segments /= 2; c.height -= 1;
            compute_bounds(P);
            c.stat_shrinks += 1; c.stat_rebalances += 1; c.stat_window_slots += c.capacity;
            root_rebalance(P, old_cap, c.capacity, c.y_m, false);
            break;
        }
        case SEQ_Y_TABLE_GROW:
            ensure_tables(P, c.table_len + 1);
            break;
        case SEQ_Y_APPEND_RUN: {
            if (dev_env("DSA_DBG_RUN") && c.dbg[4])
                fprintf(stderr, "[previous append run] ops=%lld slow=%lld fast=%.1fus slow=%.1fus shader clock %.0f MHz | model v2: entries %lld ops %lld wide events %lld "
                        "pattern misses %lld exits [end %lld, word full %lld, word empty %lld, wider level %lld]\n", (long long)c.dbg[4],
                        (long long)c.dbg[0], c.dbg[2] / 100.0, c.dbg[3] / 100.0, c.dbg[5] ? 100.0 * c.dbg[1] / c.dbg[5] : 0.0,
                        (long long)c.prof[8], (long long)c.prof[9], (long long)c.prof[10], (long long)c.prof[11], (long long)c.prof[12], (long long)c.prof[13],
                        (long long)c.prof[14], (long long)c.prof[15]);
            if (dev_env("DSA_DBG_RUN") && c.dbg[4])
                fprintf(stderr, "    model v2: %lld in-word ops simulated one by one; %lld epoch jumps; %lld wide events computed (not memoised) in %.1f us; whole model %.1f us (shader clock)\n", (long long)c.prof[3], (long long)c.prof[4],
                        (long long)c.prof[5], c.prof[6] / 2400.0, c.prof[7] / 2400.0);
            // save the bitmap, replay the run on the live bitmap, move the cells; all stream-ordered, no host wait.  The
            // device control block is authoritative afterwards (next_op, nb_elements, tables, statistics): no upload on relaunch.
            const int64_t words = (c.capacity + 63) / 64;
            const int64_t i0 = c.y_ws, R = c.y_m, n0 = c.y_we;
            hipError_t e;
            if (P.has_cols) {
                // MappedPackedCSC run: at most R new columns; expand the ops into the cell stream (semaphore cells included)
                ensure_tables(P, c.table_len + R + 1);
                if (2 * R + 1024 > P.run_cap) {
                    if (P.run_cells) hipFree(P.run_cells);
                    if (P.run_flags) hipFree(P.run_flags);
                    P.run_cap = std::max<int64_t>(2 * R + 1024, 1 << 16);
                    HIPCHK(hipMalloc(&P.run_cells, (size_t)P.run_cap * sizeof(Op)));
                    HIPCHK(hipMalloc(&P.run_flags, (size_t)(P.run_cap / 64 + 32) * sizeof(uint64_t)));
                }
                if (!P.run_out) HIPCHK(hipMalloc(&P.run_out, 2 * sizeof(int64_t)));
                HIPCHK(hipMemcpyAsync(P.d_ctl, P.h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, P.stream));     // table_cap may have grown
                e = launch_run_expand(P.d_ops, i0, R, P.d_ctl, P.col_keys, P.col_live, P.run_cells, P.run_flags, P.run_out, P.stream);
                if (e != hipSuccess) fail(DSA_EHIP, std::string("run expand launch: ") + hipGetErrorString(e));
            }
            HIPCHK(hipMemcpyAsync(P.occ_old, P.O(), (size_t)words * sizeof(uint64_t), hipMemcpyDeviceToDevice, P.stream));
            if (!P.run_memo) {                  // the memo of k_append_run, then the 8 result words of k_append_model3
                HIPCHK(hipMalloc(&P.run_memo, append_run_memo_bytes() + 8 * sizeof(int64_t)));
                HIPCHK(hipMemsetAsync(P.run_memo, 0, append_run_memo_bytes() + 8 * sizeof(int64_t), P.stream));
            }
            // the count-only replay first (appendmodel.hip); what it cannot take — short runs, small segments, a tail outside the last
            // leaf — and whatever it leaves is replayed per op by k_append_run.  DSA_MODEL3=0: per-op replay only (A/B, coverage)
            static const bool model3 = [] { const char* v = dev_env("DSA_MODEL3"); return !(v && v[0] == '0'); }();
            // (typed runs on segments below 16 slots are not count-only — appendmodel.hip — and runs below its minimum length do not pay:
            //  no launch for them)
            const bool m3_takes = model3 && !g_models_off.load() && R >= 512 && (P.has_cols ? c.segment_capacity >= 16 : c.segment_capacity >= 2) && c.capacity >= 65536;
            // typed runs on 8-slot segments (a matrix grown from the empty one: BASELINE config 5) are not count-only; their replay is the
            // per-epoch model of appendmodel.hip (k_append_model5).  DSA_MODEL5=0: per-op replay only (A/B, coverage)
            static const bool model5 = [] { const char* v = dev_env("DSA_MODEL5"); return !(v && v[0] == '0'); }();
            const bool m5_takes = model5 && !g_models_off.load() && !m3_takes && P.has_cols && c.segment_capacity == 8 && R >= 64 && c.capacity >= 256;
            int64_t* m3_out = (m3_takes || m5_takes) ? reinterpret_cast<int64_t*>(reinterpret_cast<char*>(P.run_memo) + append_run_memo_bytes()) : nullptr;
            if (m3_takes) {
                e = launch_append_model3(P.O(), P.d_ctl, R, P.has_cols ? P.run_flags : nullptr, P.has_cols ? P.run_out : nullptr, m3_out, P.stream);
            } else if (m5_takes) {
                e = launch_append_model5(P.O(), P.d_ctl, R, P.run_flags, P.run_out, m3_out, P.stream);
            }
            if ((m3_takes || m5_takes) && e != hipSuccess) {
                // The models need 140 KB of LDS per workgroup (gfx950 has 160): on a part that refuses the launch (hipFuncSetAttribute /
                // launch error) the run is not lost — the per-op replay takes all of it, and the models stay off for the process.  Only the
                // codes such a refusal produces are taken that way (and said once on stderr: config 5 is several times slower without the
                // models); anything else — a sticky error of earlier work on the stream, out of memory — is a failure like everywhere else.
                (void)hipGetLastError();
                if (e != hipErrorInvalidValue && e != hipErrorLaunchOutOfResources && e != hipErrorInvalidConfiguration && e != hipErrorSharedObjectInitFailed)
                    fail(DSA_EHIP, std::string("append model launch: ") + hipGetErrorString(e));
                if (!g_models_off.exchange(true))
                    fprintf(stderr, "libdsa_hip: append-replay models disabled for this process (%s): per-op replay from now on\n", hipGetErrorString(e));
                m3_out = nullptr;
            }
            e = launch_append_run(P.O(), P.d_ctl, i0, R, P.has_cols ? P.run_flags : nullptr, P.has_cols ? P.run_out : nullptr, P.run_memo, m3_out, P.stream);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("append run launch: ") + hipGetErrorString(e));
            permute_run(P, P.has_cols ? P.run_cells : P.d_ops, P.has_cols ? 0 : i0, n0);
            if (m3_out != nullptr && dev_env("DSA_DBG_RUN")) {
                int64_t o[8];
                HIPCHK(hipMemcpyAsync(o, m3_out, sizeof(o), hipMemcpyDeviceToHost, P.stream));
                HIPCHK(hipStreamSynchronize(P.stream));
                fprintf(stderr, "[append model %s] run of %lld ops: placed %lld status %lld reason %lld | events above the tables %lld, table levels %lld | counts %.1f us tables %.1f us driver %.1f us\n",
                        m5_takes ? "v5 (typed epochs)" : "v3", (long long)R, (long long)o[0], (long long)o[1], (long long)o[2], (long long)o[3], (long long)o[4], o[5] / 100.0, o[6] / 100.0, o[7] / 100.0);
            }
            if (++r.guard > 4 * r.n + 1000000) fail(DSA_EASSERT, "sequencer made no progress");
            seq_launch(r, false);
            return true;
        }
        default:
            fail(DSA_EASSERT, "unknown sequencer status");
    }
    if (++r.guard > 4 * r.n + 1000000) fail(DSA_EASSERT, "sequencer made no progress");
    seq_launch(r);
    return true;
}

// Runs `ops` in order on the device.  Returns the number of ops fully applied; *err receives the
// status of the failing op (0 if all were applied).
int64_t run_ops(Pma& P, const std::vector<Op>& ops, int32_t* err) {
    SeqRun r;
    seq_start(r, P, ops);
    while (seq_step(r)) {}
    *err = r.err;
    return r.applied;
}

// Batch-parallel execution of vector writes (parbatch.hip): rounds of plan / resolve / apply for the prefix of ops whose
// footprints are pairwise disjoint; the op that cuts a short prefix (and a growing chunk after it while prefixes stay
// short: ascending appends, hammering one key) goes through the sequential sequencer.  Same final state as run_ops.
int64_t run_ops_parallel(Pma& P, const OpBatch& ops, int32_t* err, bool can_fail = false) {
    *err = 0;
    const int64_t n = ops.n;
    if (n == 0) return 0;
    constexpr int GMAX = ROUND_GMAX, MIN_PREFIX = 4, ROUNDS_PER_SYNC = 12, ROUNDS_SHORT = 3;
    constexpr int64_t MERGE_AT = 256;       // pending table entries (of at most 1024) that trigger the grid-wide merge between launches
    ensure_key_width(P, ops);
    ensure_ops(P, n);
    {
        const auto tu0 = std::chrono::steady_clock::now();
        upload_batch(P, ops);
        enqueue_op_breaks(P, n);
        static const bool dbg_up = dev_env("DSA_DBG_SPLIT") != nullptr;
        if (dbg_up) fprintf(stderr, "  [run_ops_parallel] upload of %lld ops (%.1f MB, pageable): %.3f ms on the host\n", (long long)n, n * sizeof(Op) / 1e6,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count());
    }
    if (!P.d_plans) {
#ifdef DSA_FP_CHECK
        HIPCHK(hipMalloc(&P.d_plans, (size_t)GMAX * (sizeof(Plan) + FP_BYTES_PER_OP)));      // + the recorded sets of the footprint check (parbatch.hip)
        HIPCHK(hipMemsetAsync(P.d_plans, 0, (size_t)GMAX * (sizeof(Plan) + FP_BYTES_PER_OP), P.stream));
#else
        HIPCHK(hipMalloc(&P.d_plans, (size_t)GMAX * sizeof(Plan)));
#endif
        HIPCHK(hipMalloc(&P.d_pend, (size_t)2 * GMAX * sizeof(PendOp)));
        HIPCHK(hipMemsetAsync(P.d_pend, 0, (size_t)2 * GMAX * sizeof(PendOp), P.stream));
        HIPCHK(hipMalloc(&P.d_bufs, sizeof(DevBufs)));
        HIPCHK(hipHostMalloc(&P.h_bufs, sizeof(DevBufs), hipHostMallocDefault));
        std::memset(P.h_bufs, 0, sizeof(DevBufs));
        HIPCHK(hipMalloc(&P.d_rs, sizeof(RoundState)));
        HIPCHK(hipHostMalloc(&P.h_rs, sizeof(RoundState), hipHostMallocDefault));
    }
    P.h_ctl->next_op = 0; P.h_ctl->status = 0; P.h_ctl->err = 0; P.h_ctl->no_run_at = -1;
    upload_ctl(P);
    static const int64_t SEQ_CHUNK0 = [] { const char* e = dev_env("DSA_SEQ_CHUNK"); return e ? (int64_t)atoi(e) : (int64_t)8; }();
    static const int64_t BARRIER_CHUNK0 = [] { const char* e = dev_env("DSA_BARRIER_CHUNK"); return e ? (int64_t)atoi(e) : (int64_t)1; }();
    int64_t i = 0, seq_chunk = SEQ_CHUNK0;
    // run-ahead (parbatch.hip): a round applies every op that conflicts with no earlier one, the deferred ones wait in a pending list in
    // front of the fresh ops.  Only where no op can fail (a failing op must find exactly the ops in front of it applied): no tombstones,
    // not the cut batches of the tombstone path.  DSA_RUN_AHEAD=0: the prefix rule of rounds 2-5 (A/B).
    static const bool run_ahead_on = [] { const char* e = dev_env("DSA_RUN_AHEAD"); return !(e && e[0] == '0'); }();
    const bool run_ahead = run_ahead_on && !can_fail && (!P.has_cols || P.h_ctl->nb_partitions == P.h_ctl->table_len);
    int np = 0, cur = 0;                    // pending ops of the rounds and which half of d_pend holds them
    bool drain = false;                     // the next bursts work on the pending list alone ...
    int after_drain = 0;                    // ... and then: 1 switch to the local rounds, 2 the sequencer takes the chunk at the cursor
    int G = 256;
    int ema = 16 * 16;                      // RoundState::ema, carried across the bursts of the batch
    // local rounds (parbatch.hip: k_local_rounds) while the prefixes are short; a small array starts with them
    static const bool local_ok = [] { const char* e = dev_env("DSA_LOCAL_ROUNDS"); return !(e && e[0] == '0'); }();
    constexpr int LOCAL_ROUNDS = 2048, LOCAL_BELOW = 6;
    bool use_local = local_ok && (P.h_ctl->capacity <= (1 << 16) || n <= 64);      // (a handful of ops: one launch of the persistent workgroup, not a burst graph)
    // a burst that stops in its first rounds (short conflict-free prefix, barrier op) leaves the rest of its graph as no-op
    // launches (~2.5 us each, four per round): after such a stop the next burst is a short one, until one runs to its end
    int burst_rounds = ROUNDS_PER_SYNC;
    static const bool dbg_split = dev_env("DSA_DBG_SPLIT") != nullptr;
    double t_burst = 0, t_seq = 0, t_local = 0; int64_t n_burst = 0, n_seq = 0, n_yield = 0, n_local = 0, r_local = 0, o_local = 0;
    int64_t dbg_detour[32] = {0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // A batch that starts like an append run — ascending keys (vector) / ascending (column, row) pairs (MappedPackedCSC) — goes to the
    // sequencer first, which detects the run (or, when the keys are not above the last cell after all, applies a few ops and hands
    // back to the rounds): a burst of rounds on ascending appends plans and applies one op per round (0.37 ms for nothing at 100 k ops)
    bool seq_first = false;
    if (g_append_runs && P.occ_old != nullptr && n >= 64 && (!P.has_sems || P.has_cols)) {
        const int64_t probe = std::min<int64_t>(n, 256);
        seq_first = true;
        for (int64_t j = 0; j < probe && seq_first; ++j) {
            const Op o = ops.at(j);
            if (o.v == 0.0 || o.kind != (P.has_cols ? OP_MPCSC_SET : OP_VEC_SET)) seq_first = false;
            else if (j > 0) {
                const Op q = ops.at(j - 1);
                seq_first = P.has_cols ? (o.b > q.b || (o.b == q.b && o.a > q.a)) : o.a > q.a;
            }
        }
    }
    while (i < n || np > 0) {
        const auto tb0 = now();
        bool to_sequencer = seq_first;
        if (!seq_first) {
        // ---- a burst of rounds driven by the device-resident cursor; one host synchronisation per burst
        RoundState& rs = *P.h_rs;
        std::memset(&rs, 0, sizeof(rs));
        static const int tight = [] { const char* e = dev_env("DSA_TIGHT"); return e ? atoi(e) : 3; }();
        rs.cursor = i; rs.limit = n; rs.G = G; rs.min_prefix = MIN_PREFIX; rs.ema = ema; rs.tight = tight;
        rs.cursor_n = i; rs.np = rs.np_n = np; rs.cur = rs.cur_n = cur; rs.run_ahead = run_ahead ? 1 : 0; rs.drain = drain ? 1 : 0; rs.pend0 = i;
#ifdef DSA_FP_CHECK
        {   // the footprint-check build: DSA_FP_MODE = 1 recorded read / touch sets (default), 2 sequential shadow re-plan, 0 neither
            static const int fp_mode = [] { const char* e = dev_env("DSA_FP_MODE"); return e ? atoi(e) : 1; }();
            rs.tight |= fp_mode == 2 ? FP_MODE_SHADOW : (fp_mode == 1 ? FP_MODE_SETS : 0);
        }
#endif
        // the burst hands its result back through pinned memory (k_publish) and the host polls for the burst number; DSA_PUBLISH=0: two
        // device-to-host copies and a stream synchronisation instead
        const bool publish = publish_enabled();
        rs.seq = (int32_t)next_publish_seq(P);
        const BurstPublish pub = publish ? BurstPublish{P.h_rs, P.h_ctl, P.h_pub} : BurstPublish{nullptr, nullptr, nullptr};
        HIPCHK(hipMemcpyAsync(P.d_rs, P.h_rs, sizeof(RoundState), hipMemcpyHostToDevice, P.stream));
        {
            ++P.layout_epoch;
            // (every burst is followed by a stream wait, so the pinned mirror is never rewritten under a copy in flight)
            const DevBufs bufs_now{P.K().p, P.V(), P.O(), P.has_sems ? P.sems : nullptr, P.has_cols ? P.col_keys : nullptr,
                                   P.has_cols ? P.col_live : nullptr, P.wide ? 1 : 0, 0, P.d_pend};
            if (std::memcmp(&bufs_now, P.h_bufs, sizeof(DevBufs)) != 0) {
                *P.h_bufs = bufs_now;
                HIPCHK(hipMemcpyAsync(P.d_bufs, P.h_bufs, sizeof(DevBufs), hipMemcpyHostToDevice, P.stream));
            }
            // short conflict-free prefixes (a small array, colliding ops): the rounds of one persistent workgroup, no launch per round
            hipError_t e = use_local ? launch_local_rounds(P.d_bufs, P.d_ctl, P.d_ops, P.d_rs, LOCAL_ROUNDS, pub, P.stream)
                                     : launch_burst(P.d_bufs, P.d_ctl, P.d_ops, P.d_rs, P.d_plans,
                                                    burst_rounds, burst_rounds == ROUNDS_PER_SYNC ? &P.burst : &P.burst_short, pub, P.stream);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("burst launch: ") + hipGetErrorString(e));
        }
        if (publish) {
            wait_published(P);
        } else {
            HIPCHK(hipMemcpyAsync(P.h_rs, P.d_rs, sizeof(RoundState), hipMemcpyDeviceToHost, P.stream));
            HIPCHK(hipMemcpyAsync(P.h_ctl, P.d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost, P.stream));      // table_len, n_pending, counts of the burst
            HIPCHK(hipStreamSynchronize(P.stream));
        }
        t_burst += ms(tb0, now()); ++n_burst;
        if (use_local) { t_local += ms(tb0, now()); ++n_local; r_local += rs.rounds; o_local += rs.par_ops; }
        // the prefix of the last round of the burst has been applied but is folded into the cursor only by the next round's resolve step
        if (rs.pad >= 10) fail(DSA_EASSERT, "DSA_FP_CHECK: a round of the batch-parallel writes is not equivalent to the sequential order (code " + std::to_string(rs.pad) + ", details on stdout)");
        if (rs.pad == 9) fail(DSA_EASSERT, "batch-parallel writes: a deferred op left the zone it was sealed in (internal invariant of the run-ahead rounds)");
        if (rs.pad != 0) fail(DSA_EASSERT, "batch-parallel column creation left its footprint (internal invariant)");
        static const bool dbg_burst = dev_env("DSA_DBG_BURST") != nullptr;
        if (dbg_burst)
            fprintf(stderr, "    burst%s: rounds %lld ops %lld (+ last prefix %d) stop %d G %d ema %.1f pending %lld table %lld/%lld cap %lld\n", use_local ? " (local)" : "",
                    (long long)rs.rounds, (long long)rs.par_ops, rs.d, rs.stop, rs.G, rs.ema / 16.0, (long long)P.h_ctl->n_pending,
                    (long long)P.h_ctl->table_len, (long long)P.h_ctl->table_cap, (long long)P.h_ctl->capacity);
        // what the sequencer takes after a stop: the op that cannot be planned alone when the rounds were otherwise making progress
        // (the ops behind it are cheaper in a round: ~1 us each against 5-15 us), a chunk of SEQ_CHUNK0 ops when short prefixes
        // stopped them (the ops around the cursor collide); doubled while the rounds apply fewer than two ops each
        if (rs.par_ops >= 2 * std::max<int64_t>(1, rs.rounds)) seq_chunk = rs.why[7] > 0 ? SEQ_CHUNK0 : BARRIER_CHUNK0;
        // new partitions of the rounds sit at the end of the tables: back into key order with the whole chip once enough have piled up
        if (P.h_ctl->n_pending >= MERGE_AT) merge_tables(P);
        P.stat_par_rounds += rs.rounds; P.stat_par_ops += rs.par_ops; P.stat_deferred += rs.deferred;
        for (int q = 0; q < 8; ++q) P.stat_why[q] += rs.why[q];
        i = rs.cursor_n; np = rs.np_n; cur = rs.cur_n;
        G = rs.G; ema = rs.ema;
        burst_rounds = (rs.stop == 1 && rs.rounds <= ROUNDS_SHORT) ? ROUNDS_SHORT : ROUNDS_PER_SYNC;
        if (rs.stop == 5 || (drain && np == 0)) {          // the pending list is drained: what it was drained for
            drain = false;
            const int what = after_drain; after_drain = 0;
            if (what == 1) { use_local = true; continue; }
            if (what != 2) continue;
            to_sequencer = true;
        } else if (rs.stop == 1 && np > 0) {
            // the op at the head of the pending list cannot be planned (it needs the sequencer: a wide window, _extend!): everything in
            // front of it has been applied, so the sequencer takes exactly that op; then it leaves the list
            const int64_t op0 = rs.pend0;
            SeqRun r;
            r.P = &P; r.n = op0 + 1; r.n_avail = op0 + 1; r.active = true; r.defer_merge = true;
            P.h_ctl->next_op = op0; P.h_ctl->status = 0; P.h_ctl->err = 0; P.h_ctl->no_run_at = op0;       // (no append run from a pending op: the ops behind it are not its successors)
            seq_launch(r);
            while (seq_step(r)) ++n_yield;
            ++n_seq;
            if (r.err) fail(DSA_EASSERT, "batch-parallel writes: a deferred op failed in the sequencer (no op of a run-ahead batch can fail)");
            P.stat_seq_ops += 1; P.stat_seq_launches += 1;
            std::vector<PendOp> lst((size_t)np);
            HIPCHK(hipMemcpyAsync(lst.data(), P.d_pend + (size_t)cur * GMAX, (size_t)np * sizeof(PendOp), hipMemcpyDeviceToHost, P.stream));
            HIPCHK(hipStreamSynchronize(P.stream));
            if (lst[0].op != op0) fail(DSA_EASSERT, "batch-parallel writes: pending list out of step with the round state");
            --np;
            if (np > 0) { HIPCHK(hipMemcpyAsync(P.d_pend + (size_t)cur * GMAX, lst.data() + 1, (size_t)np * sizeof(PendOp), hipMemcpyHostToDevice, P.stream)); HIPCHK(hipStreamSynchronize(P.stream)); }
            continue;
        } else {
            bool want_local = false;
            if (use_local) { if (rs.stop == 3) { use_local = false; ema = 16 * 64; G = 64; } }   // full prefixes: the grid rounds pay again
            else if (local_ok && rs.rounds > 0 && ema < 16 * LOCAL_BELOW) want_local = true;     // prefixes of a few ops: one workgroup is enough
            // (the local rounds and the sequencer work on the contiguous rest of the batch: the pending list is drained first)
            if (want_local) { if (np > 0) { drain = true; after_drain = 1; continue; } use_local = true; }
            if (rs.stop != 1) continue;                       // burst used up (0), batch finished (2), or a switch of round kind (3)
            to_sequencer = true;
        }
        }
        if (!to_sequencer) continue;
        if (np > 0) { drain = true; after_drain = 2; continue; }
        seq_first = false;
        // ---- short prefix at op i: sequential sequencer for ops [i, i + seq_chunk)
        const auto ts0 = now();
        int64_t no_run_at = -1;
        for (;;) {
            SeqRun r;
            r.P = &P; r.n = std::min<int64_t>(n, i + seq_chunk); r.n_avail = n; r.active = true; r.defer_merge = true;
            P.h_ctl->next_op = i; P.h_ctl->status = 0; P.h_ctl->err = 0; P.h_ctl->no_run_at = no_run_at;
            const int64_t dbg_slots0 = P.h_ctl->stat_window_slots, dbg_reb0 = P.h_ctl->stat_rebalances, dbg_ext0 = P.h_ctl->stat_extends;
            seq_launch(r);
            while (seq_step(r)) ++n_yield;
            ++n_seq;
            if (r.err) { if (P.h_ctl->n_pending > 0) merge_tables(P); *err = r.err; return r.applied; }
            if (P.h_ctl->n_pending >= MERGE_AT) merge_tables(P);
            if (dbg_split && r.applied - i <= 2) {      // dev: what a one-op detour through the sequencer rebalanced (slots of its windows, log2 buckets)
                const int64_t ds = P.h_ctl->stat_window_slots - dbg_slots0;
                int b = 0; while ((1ll << b) < ds && b < 31) ++b;
                dbg_detour[P.h_ctl->stat_extends != dbg_ext0 ? 31 : b] += 1;
                (void)dbg_reb0;
            }
            P.stat_seq_ops += r.applied - i; P.stat_seq_launches += 1;
            i = r.applied;
            // an append run that stopped in front of op i (it needs _extend!): that op and what follows stay with the sequencer, which
            // detects the rest of the run behind it — no detour through a burst of rounds that cannot plan the op either
            if (i < n && P.h_ctl->no_run_at == i) { no_run_at = i; seq_chunk = std::max<int64_t>(seq_chunk, 8); continue; }
            break;
        }
        t_seq += ms(ts0, now());
        seq_chunk = std::min<int64_t>(seq_chunk * 2, 8192);
        G = 64;
    }
    if (dbg_split)
        fprintf(stderr, "  [run_ops_parallel %s] n=%lld: %lld bursts %.2f ms (of which %lld local launches %.2f ms: %lld mini-rounds, %lld ops), %lld sequencer chunks (%lld yields) %.2f ms\n",
                P.has_cols ? "pcsc" : "vec", (long long)n, (long long)n_burst, t_burst, (long long)n_local, t_local, (long long)r_local, (long long)o_local,
                (long long)n_seq, (long long)n_yield, t_seq);
    if (dbg_split) {
        fprintf(stderr, "    one-op detours by window slots (log2 bucket: count; 31 = with _extend!):");
        for (int b = 0; b < 32; ++b) if (dbg_detour[b]) fprintf(stderr, " %d:%lld", b, (long long)dbg_detour[b]);
        fprintf(stderr, "\n");
        fprintf(stderr, "    sequencer chunks: waiting for the device %.2f ms; host work by yield kind [done %.2f, rebalance %.2f, extend %.2f, shrink %.2f, table %.2f, error %.2f, run %.2f] ms\n",
                g_seq_wait_ms, g_seq_host_ms[0], g_seq_host_ms[1], g_seq_host_ms[2], g_seq_host_ms[3], g_seq_host_ms[4], g_seq_host_ms[5], g_seq_host_ms[6]);
        fprintf(stderr, "    seq_launch calls %.2f ms\n", g_seq_launch_ms);
        g_seq_wait_ms = 0; g_seq_launch_ms = 0; for (double& x : g_seq_host_ms) x = 0;
    }
    if (P.h_ctl->n_pending > 0) merge_tables(P);          // the tables leave the batch in key order (the reference's numbering)
    return n;
}

// Two independent structures (the colmajor and rowmajor orientation): both sequencers run at the same time, each on
// its own stream; the host alternates between their yield mailboxes.
void run_ops_pair(Pma& A, const std::vector<Op>& opsA, Pma& B, const std::vector<Op>& opsB, SeqRun& ra, SeqRun& rb) {
    seq_start(ra, A, opsA);
    seq_start(rb, B, opsB);
    while (ra.active || rb.active) {
        if (ra.active) seq_step(ra);
        if (rb.active) seq_step(rb);
    }
}

void pma_info(Pma& P, int64_t nb_partitions_or_len, int64_t* info) {
    const Ctl& c = *P.h_ctl;
    std::memset(info, 0, sizeof(int64_t) * DSA_INFO_COUNT);
    info[DSA_INFO_CAPACITY] = c.capacity;
    info[DSA_INFO_SEGMENT_CAPACITY] = c.segment_capacity;
    info[DSA_INFO_NB_SEGMENTS] = c.nb_segments;
    info[DSA_INFO_NB_ELEMENTS] = c.nb_elements;
    info[DSA_INFO_HEIGHT] = c.height;
    info[DSA_INFO_NB_PARTITIONS] = nb_partitions_or_len;
    info[DSA_INFO_TABLE_LEN] = c.table_len;
    info[DSA_INFO_STAT_WINDOW_SLOTS] = c.stat_window_slots;
    info[DSA_INFO_STAT_REBALANCES] = c.stat_rebalances;
    info[DSA_INFO_STAT_EXTENDS] = c.stat_extends;
    info[DSA_INFO_STAT_SHRINKS] = c.stat_shrinks;
    info[11] = P.stat_par_rounds; info[12] = P.stat_par_ops; info[13] = P.stat_seq_ops;
    info[DSA_INFO_STAT_SPMV_NOMEMSET] = P.stat_spmv_nomemset;
    info[DSA_INFO_STAT_GRID_REBALANCES] = P.stat_grid_rebalances;
    // HBM held by the structure: both slot buffers (keys, values, bitmap), the saved bitmap of append runs, the tables and the merge scratch
    info[DSA_INFO_HBM_BYTES] = 2 * (P.cap_alloc * (int64_t)(P.kb() + sizeof(double)) + P.occ_words * 8) + (P.occ_old ? P.occ_words * 8 : 0) +
                               (P.has_sems ? c.table_cap * 8 : 0) + (P.has_cols ? c.table_cap * 9 : 0) + 2 * P.tmerge_cap * 8 +
                               (P.d_ops ? P.ops_cap * (int64_t)sizeof(Op) + (P.ops_cap / 64 + 8) * 8 : 0) + (P.d_opsrc ? P.opsrc_cap * 24 : 0);      // op array, run-break bitmap, batch columns
}

void export_slots(Pma& P, int64_t* keys, double* vals, uint8_t* occ, int64_t cap) {
    const int64_t c = P.capacity();
    if (cap < c) fail(DSA_ECAP, "output buffers smaller than capacity");
    std::vector<uint64_t> words((size_t)((c + 63) / 64));
    download_keys(P, keys, P.keys[P.cur], c);
    HIPCHK(hipMemcpyAsync(vals, P.V(), (size_t)c * sizeof(double), hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipMemcpyAsync(words.data(), P.O(), words.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    for (int64_t i = 0; i < c; ++i) {
        const uint8_t o = (words[(size_t)(i >> 6)] >> (i & 63)) & 1ull;
        occ[i] = o;
        if (!o) { keys[i] = 0; vals[i] = 0.0; }
    }
}

void export_tables(Pma& P, int64_t* semaphores, int64_t* col_keys, uint8_t* col_live, int64_t table_cap) {
    const int64_t tl = P.h_ctl->table_len;
    if (table_cap < tl) fail(DSA_ECAP, "table buffers too small");
    if (tl == 0) return;
    HIPCHK(hipMemcpyAsync(semaphores, P.sems, (size_t)tl * sizeof(int64_t), hipMemcpyDeviceToHost, P.stream));
    if (col_keys) {
        HIPCHK(hipMemcpyAsync(col_keys, P.col_keys, (size_t)tl * sizeof(int64_t), hipMemcpyDeviceToHost, P.stream));
        HIPCHK(hipMemcpyAsync(col_live, P.col_live, (size_t)tl, hipMemcpyDeviceToHost, P.stream));
    }
    HIPCHK(hipStreamSynchronize(P.stream));
    if (col_keys) for (int64_t i = 0; i < tl; ++i) if (!col_live[i]) col_keys[i] = 0;
}

void pma_check(Pma& P, int64_t* report) {
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc(&d, 8 * sizeof(unsigned long long)));
    unsigned long long r[8] = {0};
    hipError_t e = launch_check(P.K(), P.V(), P.O(), P.capacity(), P.occ_words, P.has_sems ? P.sems : nullptr,
                                P.has_cols ? P.col_keys : nullptr, P.has_cols ? P.col_live : nullptr, P.h_ctl->table_len, d, P.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r, d, sizeof(r), hipMemcpyDeviceToHost, P.stream);
    // no table entry may be pending outside a batch (tables.hip): the DEVICE copy of the counter is the one the kernels trust
    int64_t dev_pending = 0, merge_fault = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&dev_pending, reinterpret_cast<const char*>(P.d_ctl) + offsetof(Ctl, n_pending), sizeof(int64_t), hipMemcpyDeviceToHost, P.stream);
    // the grid-wide table merge raises hdr[2] if it was ever handed more entries than it takes (cannot happen: TABLE_PEND_MAX)
    if (e == hipSuccess && P.tmerge.hdr != nullptr) e = hipMemcpyAsync(&merge_fault, P.tmerge.hdr + 2, sizeof(int64_t), hipMemcpyDeviceToHost, P.stream);
    unsigned long long move_fault = 0;
    if (e == hipSuccess && P.work.status != nullptr)
        e = hipMemcpyAsync(&move_fault, P.work.status + P.work.status_cap - 1, sizeof(move_fault), hipMemcpyDeviceToHost, P.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(P.stream);
    hipFree(d);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("check: ") + hipGetErrorString(e));
    if (move_fault != 0) fail(DSA_EHIP, "a rebalance launch gave up waiting for its prefix table (k_move2: dispatch-order assumption violated)");
    for (int i = 0; i < 8; ++i) report[i] = (int64_t)r[i];
    const int64_t live = P.has_sems ? P.h_ctl->nb_partitions : 0;
    report[6] = (report[0] != P.h_ctl->nb_elements || report[1] != live || dev_pending != 0 || P.h_ctl->n_pending != 0 || merge_fault != 0) ? 1 : 0;
}

void ensure_q(Pma& P, int64_t n) {
    if (n <= P.q_cap) return;
    if (P.d_q) hipFree(P.d_q);
    P.q_cap = std::max<int64_t>(n, 256);
    HIPCHK(hipMalloc(&P.d_q, (size_t)P.q_cap * 3 * sizeof(double)));
}

// batched getindex on the device; mode as in launch_get_batch
void get_batch(Pma& P, int mode, const int64_t* qa, const int64_t* qb, int64_t n, double* out) {
    if (n <= 0) return;
    if (n <= 64 && publish_enabled()) {
        // a scalar getindex or a handful of them: one launch that reads its queries from, and writes its answers to, pinned memory
        for (int64_t i = 0; i < n; ++i) { P.h_get[i] = qa[i]; P.h_get[64 + i] = qb ? qb[i] : 0; }
        const unsigned long long seq = ++P.get_seq;
        __atomic_thread_fence(__ATOMIC_RELEASE);
        hipError_t e = launch_get_small(mode, P.K(), P.V(), P.O(), P.capacity(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len, P.h_get, (int)n, seq, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("get launch: ") + hipGetErrorString(e));
        wait_policy_block(P);
        volatile int64_t* seqp = P.h_get + 193;
        auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        while ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) {
            if (std::chrono::steady_clock::now() < next_query) continue;
            const hipError_t q = hipStreamQuery(P.stream);
            if (q == hipErrorNotReady) { next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2); continue; }
            if (q != hipSuccess) fail(DSA_EHIP, std::string("get: ") + hipGetErrorString(q));
            if ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) fail(DSA_EHIP, "lookup kernel finished without publishing its result");
        }
        std::memcpy(out, P.h_get + 128, (size_t)n * sizeof(double));
        const int32_t err = (int32_t)P.h_get[192];
        if (err) fail(err, err == DSA_EBOUNDS ? "partition index out of range" : "partition has no semaphore");
        return;
    }
    ensure_q(P, n);
    int64_t* d_qa = reinterpret_cast<int64_t*>(P.d_q);
    int64_t* d_qb = d_qa + P.q_cap;
    double* d_out = P.d_q + 2 * P.q_cap;
    HIPCHK(hipMemcpyAsync(d_qa, qa, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, P.stream));
    if (qb) HIPCHK(hipMemcpyAsync(d_qb, qb, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, P.stream));
    HIPCHK(hipMemsetAsync(P.d_err, 0, sizeof(int32_t), P.stream));
    hipError_t e = launch_get_batch(mode, P.K(), P.V(), P.O(), P.capacity(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len,
                                    d_qa, d_qb, n, d_out, P.d_err, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("get launch: ") + hipGetErrorString(e));
    int32_t err = 0;
    HIPCHK(hipMemcpyAsync(out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipMemcpyAsync(&err, P.d_err, sizeof(int32_t), hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    if (err) fail(err, err == DSA_EBOUNDS ? "partition index out of range" : "partition has no semaphore");
}

// stored cells of the slot range [from, to] in slot order: K-pack on the device into the alternate buffer (free between
// rebalances), then only the packed cells cross PCIe
// the pinned landing area of views, small packs and small builds: 8 header words (meta [0..4], sequence number [5]) + 2 x 1024 cells
constexpr int64_t VIEW_AREA_CELLS = 1024;
void ensure_view_area(Pma& P) {
    if (P.h_view) return;
    HIPCHK(pinned_alloc(reinterpret_cast<void**>(&P.h_view), (size_t)(8 + 2 * VIEW_AREA_CELLS) * sizeof(int64_t)));
    std::memset(P.h_view, 0, 8 * sizeof(int64_t));          // header: a stale sequence number of the block's previous user must not match
}
// The area is LEASED for one operation and goes back to the pinned pool (pool.hip keeps idle blocks by size class: a lease costs a
// map lookup) when the operation is over: 10^5 small vectors — Coluna keeps that many — would otherwise pin 32 KB each for life.
// The sequence numbers stay per handle and start at 1; the header is zeroed at every lease.  When the operation fails with a kernel
// possibly still in flight the block stays with the handle (released with it) instead of being handed to somebody else.
struct ViewAreaLease {
    Pma& P; int exc;
    explicit ViewAreaLease(Pma& p) : P(p), exc(std::uncaught_exceptions()) { ensure_view_area(P); }
    ~ViewAreaLease() {
        if (std::uncaught_exceptions() > exc) return;
        pinned_free(P.h_view); P.h_view = nullptr;
    }
};
// polls word [5] of the landing area for `seq` (the stream is asked now and then: a failed launch cannot hang the host)
void wait_view_seq(Pma& P, unsigned long long seq, const char* what) {
    wait_policy_block(P);
    volatile int64_t* seqp = P.h_view + 5;
    auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    while ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) {
        if (std::chrono::steady_clock::now() < next_query) continue;
        const hipError_t q = hipStreamQuery(P.stream);
        if (q == hipErrorNotReady) { next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2); continue; }
        if (q != hipSuccess) fail(DSA_EHIP, std::string(what) + ": " + hipGetErrorString(q));
        if ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) fail(DSA_EHIP, std::string(what) + ": kernel finished without publishing its result");
    }
}

void view_small(Pma& P, int64_t col, int64_t range_from, int64_t range_to, std::vector<int64_t>& ks, std::vector<double>& vs);
void read_range_general(Pma& P, int64_t from, int64_t to, std::vector<int64_t>& ks, std::vector<double>& vs);
void read_range(Pma& P, int64_t from, int64_t to, std::vector<int64_t>& ks, std::vector<double>& vs) {
    // up to VIEW_SMALL_SLOTS slots (iteration over a small vector, a short slice): one launch that packs the cells and hands the first 512 to the
    // host through pinned memory (nonzeros() of a 100-entry vector: 80 -> 25 us); longer ranges: tile counts + scan + K-pack
    if (to >= from && from >= 1 && to - from + 1 <= VIEW_SMALL_SLOTS && to - from + 1 <= P.cap_alloc && publish_enabled()) { view_small(P, 0, from, to, ks, vs); return; }
    read_range_general(P, from, to, ks, vs);
}
void read_range_general(Pma& P, int64_t from, int64_t to, std::vector<int64_t>& ks, std::vector<double>& vs) {
    ks.clear(); vs.clear();
    if (to < from) return;
    const int alt = 1 - P.cur;
    int64_t cnt = 0;
    hipError_t e = launch_compact_range(P.K(), P.V(), P.O(), from, to, P.KA(alt), P.vals[alt], P.cap_alloc, &P.work, &cnt, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("compact launch: ") + hipGetErrorString(e));
    if (cnt == 0) return;
    ks.resize((size_t)cnt); vs.resize((size_t)cnt);
    HIPCHK(hipMemcpyAsync(vs.data(), P.vals[alt], (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, P.stream));
    download_keys(P, ks.data(), P.keys[alt], cnt);        // synchronises
}

// ------------------------------------------------------------------------------------------------
// bulk builders (K-build): everything but the staging of the caller's arrays runs on the device
// ------------------------------------------------------------------------------------------------
// order of the emit + spread phases of concurrent K-builds on one device (see pma_build_dev)
constexpr int MAX_EMIT_DEVICES = 16;
static std::mutex g_emit_mu;
static hipEvent_t g_emit_done[MAX_EMIT_DEVICES] = {};
// K-build from device-resident triples: sort / combine / emit on the device (build.hip), then the full-array spread;
// semaphores[] positions are written by the spread kernel.
//   mode 0: one orientation of a matrix (MappedPackedCSC: partitions = distinct values of d_part)
//   mode 1: a vector (d_part == nullptr, no semaphores)         dynamicsparsevec  src/vector.jl:38-62
//   mode 2: PackedCSC with explicit partition ids 1..nparts      PackedCSC ctor    src/pcsr.jl:26-63
void pma_build_dev(Pma& P, const int64_t* d_part, const int64_t* d_key, const double* d_val, int64_t nnz, int32_t combine,
                   int mode, int64_t nparts_explicit, bool wide, KeyRange part_range = KeyRange(), KeyRange key_range = KeyRange()) {
    P.wide = wide;                     // decided by the caller from the host copy of the keys, before anything is allocated
    // fault injection for the error paths of the builders (tests): DSA_FAIL_BUILD=1 fails every build while it is set
    if (const char* fe = dev_env("DSA_FAIL_BUILD")) if (fe[0] == '1') fail(DSA_EHIP, "injected build failure (DSA_FAIL_BUILD)");
    if (nnz == 0) {
        std::vector<int64_t> ks; std::vector<double> vs;
        const int64_t np = mode == 2 ? nparts_explicit : 0;
        for (int64_t p = 1; p <= np; ++p) { ks.push_back(SEM_KEY); vs.push_back((double)p); }    // only semaphore cells
        if (P.has_sems) { P.h_ctl->nb_partitions = np; P.h_ctl->table_len = np; ensure_tables(P, std::max<int64_t>(2 * np, 64)); }
        build_from_packed(P, ks, vs);
        return;
    }
    BuildScratch sc;
    int64_t counts[2] = {0, 0};
    static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    // While the sort kernels run (build_prepare calls this between enqueueing them and waiting for the counts): tables and slot buffers
    // for the UPPER bounds — every triple a cell of its own, every partition of the key range present.  The exact sizes are known only
    // from the counts, but capacity_for is monotone and the allocations (13 of them, a dozen memsets) used to sit between the sort and
    // the emit with the GPU idle: 180 of the 1500 us of config 3's closefillmode!.  Duplicates folded later only leave the buffers
    // larger than needed (as after a _shrink!).  Only for a structure that holds nothing yet: growing an existing one waits for
    // the stream (old contents are copied).
    // BEST EFFORT: the upper bound can be far above what the counts will ask for (duplicate-heavy input: a fill buffer that overwrites
    // the same cells, a vector fed repeated keys), so it is capped at a share of the memory that is free right now, an allocation that
    // fails here is undone (the exact sizing below gets its chance), and buffers more than 4 x too large are handed back once the
    // counts are known.
    bool prealloc_done = false;
    auto release_prealloc = [&] {
        (void)hipStreamSynchronize(P.stream);      // (the counts arrive through pinned memory: the memsets of the speculative blocks may still be queued)
        pma_free_buffers(P);
        P.cap_alloc = 0; P.occ_words = 0; P.occ_dirty[0] = P.occ_dirty[1] = 0;
        pool_free(P.sems); pool_free(P.col_keys); pool_free(P.col_live);
        P.sems = nullptr; P.col_keys = nullptr; P.col_live = nullptr; P.h_ctl->table_cap = 0;
        prealloc_done = false;
    };
    const std::function<void()> prealloc = [&] {
        if (P.cap_alloc != 0 || P.sems != nullptr) return;
        int64_t np_ub = mode == 2 ? nparts_explicit : 0;
        if (mode == 0) np_ub = part_range.known() ? std::min<int64_t>(nnz, (int64_t)std::min<uint64_t>((uint64_t)part_range.hi - (uint64_t)part_range.lo, (uint64_t)nnz) + 1) : nnz;
        const int64_t slots_ub = 2 * capacity_for(nnz + np_ub);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
        const double want = 2.0 * (double)slots_ub * (double)(P.kb() + sizeof(double)) + 4.0 * 17.0 * (double)np_ub;
        if (want > 0.25 * ((double)free_b + (double)pool_idle_bytes())) return;          // not speculatively: exact sizing after the counts
        try {
            if (P.has_sems) ensure_tables(P, std::max<int64_t>(2 * np_ub, 64));
            ensure_capacity_alloc(P, slots_ub);
            prealloc_done = true;
        } catch (const Fail&) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(P.stream);
            release_prealloc();
        }
    };
    hipError_t e;
    try { e = build_prepare(d_part, d_key, d_val, nnz, part_range, key_range, sc, counts, P.stream, &prealloc); }
    catch (...) { build_abort(sc); throw; }          // (an allocation of `prealloc` failed: the scratch of the sort is released here)
    const auto tp1 = std::chrono::steady_clock::now();
    if (e != hipSuccess) fail(DSA_EHIP, std::string("K-build prepare: ") + hipGetErrorString(e));
    const int64_t np = mode == 0 ? counts[1] : (mode == 2 ? nparts_explicit : 0);
    const int64_t n = counts[0] + np;
    try {
        if (prealloc_done && P.cap_alloc > 8 * capacity_for(n) && P.cap_alloc > (1 << 20)) release_prealloc();
        if (P.has_sems) {
            P.h_ctl->nb_partitions = np; P.h_ctl->table_len = np;
            ensure_tables(P, std::max<int64_t>(2 * np, 64));
        }
        const int64_t capacity = capacity_for(n);
        set_geometry_for_new(P, capacity, n);
        ensure_capacity_alloc(P, 2 * capacity);
        if (P.has_cols && np > 0) HIPCHK(hipMemsetAsync(P.col_live, 1, (size_t)np, P.stream));
    } catch (...) { build_abort(sc); throw; }
    const auto tp2 = std::chrono::steady_clock::now();
    ++P.layout_epoch;
    // (the emit kernels are only enqueued: the spread goes in right behind them, the scratch of the sort is released after the one
    //  stream wait of upload_ctl instead of after a wait of its own)
    // The two orientations of a matrix are built side by side on two streams.  Their sort passes share the chip well; their emits — ten
    // million 8-byte gathers of the values by input index each — and spreads do not: 293 + 263 us side by side against 100 us each alone,
    // the spreads 118 + 78 against 54.  So the emit + spread of one build waits (on the device: an event, no host wait) for the emit +
    // spread of the build enqueued before it.
    std::unique_lock<std::mutex> emit_order(g_emit_mu);
    const bool ordered = P.device >= 0 && P.device < MAX_EMIT_DEVICES;      // (events belong to a device: one slot per device)
    if (ordered) {
        hipEvent_t& ev = g_emit_done[P.device];
        try {
            if (ev == nullptr) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            else HIPCHK(hipStreamWaitEvent(P.stream, ev, 0));
        } catch (...) { emit_order.unlock(); build_abort(sc); throw; }      // (the scratch of the sort would leak otherwise)
    }
    e = build_emit(d_val, combine, sc, P.K(), P.V(), P.has_cols ? P.col_keys : nullptr, mode, nparts_explicit, P.stream, false);
    if (e != hipSuccess) { emit_order.unlock(); build_abort(sc); fail(DSA_EHIP, std::string("K-build emit: ") + hipGetErrorString(e)); }
    const auto tp3 = std::chrono::steady_clock::now();
    P.h_ctl->stat_rebalances = 0; P.h_ctl->stat_window_slots = 0;
    if (P.capacity() != P.h_ctl->segment_capacity) { P.h_ctl->stat_rebalances = 1; P.h_ctl->stat_window_slots = P.capacity(); }
    try {
        root_rebalance(P, n, P.capacity(), n, true);
        if (ordered) HIPCHK(hipEventRecord(g_emit_done[P.device], P.stream));
        emit_order.unlock();
        upload_ctl(P);
    } catch (...) { if (emit_order.owns_lock()) emit_order.unlock(); build_abort(sc); throw; }
    build_abort(sc);          // (stream already waited for: releases the scratch)
    if (dbg_time) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "  [pma_build_dev] prepare (alloc + sorts + scans) %.1f ms  tables/slot alloc %.1f ms  emit (+ free) %.1f ms  spread + ctl %.1f ms\n",
                ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3), ms(tp3, std::chrono::steady_clock::now()));
    }
}

// Both orientations of a matrix from ONE sort of the caller's triples (round 6).  A = the orientation whose partitions are d_part
// (sorted as before: composite of (partition, key) + input index, values gathered at the emit); its emit also leaves the folded cells
// as composites of the TWIN B — partition bits and key bits swapped — which B's builder only has to sort by its partition bits
// (build_derived_*: 3 passes of 16-byte records instead of 5 passes + a composite pass, and no 10 M random gathers at its emit).
// B's sort runs on B's stream behind A's emit (an event) while A's spread is still running; A's slot buffers and tables are sized
// while A's sort runs, B's while B's.  Returns false — with nothing done to B and A built as usual — when A's composite does not fit
// 64 bits (the general path of build.hip): the caller then builds B from the triples.
bool mat_build_both_dev(Pma& A, Pma& B, const int64_t* d_part, const int64_t* d_key, const double* d_val, int64_t nnz, bool wideA, bool wideB,
                        KeyRange part_range, KeyRange key_range) {
    A.wide = wideA; B.wide = wideB;
    BuildScratch sa, sb;
    int64_t ca[2] = {0, 0}, cb[2] = {0, 0};
    bool sb_live = false;
    auto size_for = [](Pma& P, int64_t ncells, int64_t np) {
        P.h_ctl->nb_partitions = np; P.h_ctl->table_len = np;
        ensure_tables(P, std::max<int64_t>(2 * np, 64));
        const int64_t n = ncells + np;
        set_geometry_for_new(P, capacity_for(n), n);
        ensure_capacity_alloc(P, 2 * P.capacity());
        if (np > 0) HIPCHK(hipMemsetAsync(P.col_live, 1, (size_t)np, P.stream));
        ++P.layout_epoch;
        P.h_ctl->stat_rebalances = 0; P.h_ctl->stat_window_slots = 0;
        if (P.capacity() != P.h_ctl->segment_capacity) { P.h_ctl->stat_rebalances = 1; P.h_ctl->stat_window_slots = P.capacity(); }
        return n;
    };
    // best-effort sizing under the sort kernels (see pma_build_dev): the upper bounds, when a quarter of the free memory covers them
    // (the memsets of the fresh blocks — tables, bitmaps: ~60 us per orientation — go to the OTHER orientation's stream, idle at that
    //  moment, instead of queueing behind the sort on the orientation's own; its emit waits for them through an event)
    auto prealloc_for = [&](Pma& P, int64_t cells_ub, const KeyRange& pr, hipStream_t side) {
        return [&P, cells_ub, pr, side] {
            if (P.cap_alloc != 0 || P.sems != nullptr) return;
            struct Swap { Pma& P; hipStream_t own; Swap(Pma& p, hipStream_t s) : P(p), own(p.stream) { P.stream = s; }
                          ~Swap() { P.stream = own; } };
            const int64_t np_ub = pr.known() ? std::min<int64_t>(cells_ub, (int64_t)std::min<uint64_t>((uint64_t)pr.hi - (uint64_t)pr.lo, (uint64_t)cells_ub) + 1) : cells_ub;
            const int64_t slots_ub = 2 * capacity_for(cells_ub + np_ub);
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
            if (2.0 * (double)slots_ub * (double)(P.kb() + sizeof(double)) + 68.0 * (double)np_ub > 0.25 * ((double)free_b + (double)pool_idle_bytes())) return;
            try {
                {
                    Swap sw(P, side);
                    ensure_tables(P, std::max<int64_t>(2 * np_ub, 64)); ensure_capacity_alloc(P, slots_ub);
                }
                if (P.ev_handoff == nullptr) HIPCHK(hipEventCreateWithFlags(&P.ev_handoff, hipEventDisableTiming));
                HIPCHK(hipEventRecord(P.ev_handoff, side));
                HIPCHK(hipStreamWaitEvent(P.stream, P.ev_handoff, 0));
            } catch (const Fail&) {
                (void)hipGetLastError(); (void)hipStreamSynchronize(side); (void)hipStreamSynchronize(P.stream);
                pma_free_buffers(P); P.cap_alloc = 0; P.occ_words = 0; P.occ_dirty[0] = P.occ_dirty[1] = 0;
                pool_free(P.sems); pool_free(P.col_keys); pool_free(P.col_live);
                P.sems = nullptr; P.col_keys = nullptr; P.col_live = nullptr; P.h_ctl->table_cap = 0;
            }
        };
    };
    try {
        const std::function<void()> pa = prealloc_for(A, nnz, part_range, B.stream);
        hipError_t e = build_prepare(d_part, d_key, d_val, nnz, part_range, key_range, sa, ca, A.stream, &pa);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("K-build prepare: ") + hipGetErrorString(e));
        const bool derive = !sa.wide_path;
        const int64_t na = size_for(A, ca[0], ca[1]);
        if (derive) {
            e = build_derived_alloc(sb, ca[0], /*kbits*/ sa.pbits, /*pbits*/ sa.kbits, /*kmin*/ sa.pmin, /*pmin*/ sa.kmin, B.stream);
            sb_live = true;
            if (e != hipSuccess) fail(DSA_EHIP, std::string("K-build (twin) scratch: ") + hipGetErrorString(e));
        }
        e = build_emit(d_val, DSA_COMBINE_ADD, sa, A.K(), A.V(), A.col_keys, 0, 0, A.stream, false, derive ? sb.comp[0] : nullptr, derive ? sb.val[0] : nullptr);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("K-build emit: ") + hipGetErrorString(e));
        if (derive) {
            if (A.ev_handoff == nullptr) HIPCHK(hipEventCreateWithFlags(&A.ev_handoff, hipEventDisableTiming));
            HIPCHK(hipEventRecord(A.ev_handoff, A.stream));
            HIPCHK(hipStreamWaitEvent(B.stream, A.ev_handoff, 0));
        }
        root_rebalance(A, na, A.capacity(), na, true);
        if (derive) {
            // B: sort the cells A's emit left by B's partition bits (behind the event), flags, counts
            const KeyRange kb = key_range;      // B's partitions are A's keys
            const std::function<void()> pb = prealloc_for(B, ca[0], kb, A.stream);
            e = build_derived_sort(sb, cb, B.stream, &pb);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("K-build (twin) sort: ") + hipGetErrorString(e));
            if (cb[0] != ca[0]) fail(DSA_EASSERT, "K-build: the twin orientation counts other cells than its sibling emitted");
            const int64_t nb = size_for(B, cb[0], cb[1]);
            e = build_emit(nullptr, DSA_COMBINE_ADD, sb, B.K(), B.V(), B.col_keys, 0, 0, B.stream, false);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("K-build (twin) emit: ") + hipGetErrorString(e));
            root_rebalance(B, nb, B.capacity(), nb, true);
        }
        upload_ctl(A);                   // (waits for A's stream: its scratch is free)
        build_abort(sa);
        if (derive) { upload_ctl(B); build_abort(sb); sb_live = false; }
        return derive;
    } catch (...) {
        build_abort(sa);
        if (sb_live) build_abort(sb);
        throw;
    }
}

// uploads host arrays (any of them may be nullptr) and runs the device builder
void pma_build_from_host(Pma& P, const int64_t* part, const int64_t* key, const double* val, int64_t nnz, int32_t combine,
                         int mode, int64_t nparts_explicit) {
    static const bool small_build = [] { const char* e = dev_env("DSA_SMALL_BUILD"); return !(e && e[0] == '0'); }();
    if (small_build && mode == 1 && part == nullptr && nnz >= 1 && nnz <= VIEW_AREA_CELLS && publish_enabled() && dev_env("DSA_FAIL_BUILD") == nullptr) {
        // a small vector: ONE launch sorts, folds and packs the caller's pairs (read from the pinned landing area) in front of the slot
        // buffers and hands the entry count back; then the spread.  130 -> ~45 us for 50 entries (DSA_SMALL_BUILD=0: the general builder)
        KeyScan ks; ks.add(key, nnz);
        P.wide = !ks.fit32();
        ViewAreaLease lease(P);
        std::memcpy(P.h_view + 8, key, (size_t)nnz * sizeof(int64_t));
        std::memcpy(P.h_view + 8 + VIEW_AREA_CELLS, val, (size_t)nnz * sizeof(double));
        ensure_capacity_alloc(P, 2 * capacity_for(nnz));          // (an upper bound: folding can only shorten the stream)
        const unsigned long long seq = ++P.view_seq;
        __atomic_thread_fence(__ATOMIC_RELEASE);
        hipError_t e = launch_build_small_vec(P.h_view, (int)nnz, (int)VIEW_AREA_CELLS, combine, P.K(), P.V(), seq, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("small build launch: ") + hipGetErrorString(e));
        wait_view_seq(P, seq, "small build");
        const int64_t n = P.h_view[0];
        if (n < 1 || n > nnz) fail(DSA_EASSERT, "small build returned an impossible entry count");
        const int64_t capacity = capacity_for(n);
        set_geometry_for_new(P, capacity, n);
        ++P.layout_epoch;
        P.h_ctl->stat_rebalances = 0; P.h_ctl->stat_window_slots = 0;
        if (P.capacity() != P.h_ctl->segment_capacity) { P.h_ctl->stat_rebalances = 1; P.h_ctl->stat_window_slots = P.capacity(); }
        root_rebalance(P, n, P.capacity(), n, true);
        upload_ctl(P);
        return;
    }
    int64_t *dP = nullptr, *dK = nullptr; double* dV = nullptr;
    auto release = [&] { pool_free(dP); pool_free(dK); pool_free(dV); };
    try {
        if (nnz > 0) {
            if (part) { HIPCHK(pool_alloc(reinterpret_cast<void**>(&dP), (size_t)nnz * 8)); HIPCHK(hipMemcpyAsync(dP, part, (size_t)nnz * 8, hipMemcpyHostToDevice, P.stream)); }
            HIPCHK(pool_alloc(reinterpret_cast<void**>(&dK), (size_t)nnz * 8)); HIPCHK(hipMemcpyAsync(dK, key, (size_t)nnz * 8, hipMemcpyHostToDevice, P.stream));
            HIPCHK(pool_alloc(reinterpret_cast<void**>(&dV), (size_t)nnz * 8)); HIPCHK(hipMemcpyAsync(dV, val, (size_t)nnz * 8, hipMemcpyHostToDevice, P.stream));
            HIPCHK(hipStreamSynchronize(P.stream));
        }
        KeyScan ks; ks.add(key, nnz);                        // (runs while the uploads above are in flight when they are asynchronous)
        KeyRange pr;
        if (part && mode == 2) { pr.lo = 1; pr.hi = std::max<int64_t>(nparts_explicit, 1); }
        pma_build_dev(P, dP, dK, dV, nnz, combine, mode, nparts_explicit, !ks.fit32(), pr, ks.range());
    } catch (...) {
        (void)hipStreamSynchronize(P.stream);
        release();
        throw;
    }
    release();
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
// Single setindex! calls are write-combined on the host (the "buffered batched writes" of the path): they are queued and
// applied, in order, by ONE sequencer launch at the latest before the next call that observes the structure.
constexpr size_t PENDING_FLUSH = 1u << 16;
struct dsa_vec { Pma P; int64_t n = 0; std::vector<int64_t> pk; std::vector<double> pv; };
struct dsa_pcsc { Pma P; };
// Buffer  src/buffer.jl:1-4 — the fill-mode write buffer, DEVICE-RESIDENT: appended triples are staged in two pinned host chunks
// and uploaded asynchronously as a chunk fills (the copy of chunk k overlaps the caller's appends into chunk k+1), so that
// closefillmode! finds the (row, col, value) stream already in HBM and only ships the last partial chunk.  The reference keeps
// a Dict row -> (colids, vals); what it uses the per-row structure for — rejecting a second addrow! of a row id
// (src/buffer.jl:13) — is the `rows` set here; the order of the entries is irrelevant after the (col, row) sort of the
// builder, duplicates of (i, j) are accumulated with + at the flush like the reference (test/functional/sparsematrix.jl:433-437).
struct FillBuffer {
    static constexpr int64_t CHUNK = 1 << 20;            // triples per pinned staging chunk (24 MB)
    static constexpr int64_t EAGER = 1 << 16;            // a finished batch ships its staged triples at once from this many on
    std::vector<uint64_t> row_bits;                       // rows 1 .. 2^28 already written (one bit each, grown on demand)
    std::unordered_set<int64_t> rows_far;                 // ... and the others
    int64_t* hI[2] = {nullptr, nullptr}; int64_t* hJ[2] = {nullptr, nullptr}; double* hV[2] = {nullptr, nullptr};
    hipEvent_t uploaded[2] = {nullptr, nullptr};
    bool in_flight[2] = {false, false};
    static constexpr int64_t PIECE = 1 << 18;            // a chunk goes up in pieces of this many triples, behind the memcpy that stages them
    int cur = 0;
    int64_t fill = 0;                                     // triples in the current pinned chunk
    int64_t sent = 0;                                     // ... of which already on their way to HBM
    int64_t *dI = nullptr, *dJ = nullptr; double* dV = nullptr;
    int64_t dcap = 0, dlen = 0;                           // triples allocated / resident in HBM
    long long* d_acc = nullptr;                           // running value ranges of the resident triples (build.hip: k_minmax_acc), 5 words
    long long* h_acc = nullptr;                           // pinned read-back of them
    hipStream_t stream = nullptr;
    int64_t length = 0;
    int device = 0;
};
static inline bool fill_row_test_and_set(FillBuffer& b, int64_t row) {
    if (row >= 1 && row < ((int64_t)1 << 28)) {
        const size_t w = (size_t)(row >> 6);
        if (w >= b.row_bits.size()) b.row_bits.resize(std::max<size_t>(2 * b.row_bits.size(), w + 1024), 0ull);
        const uint64_t bit = 1ull << (row & 63);
        const bool was = (b.row_bits[w] & bit) != 0;
        b.row_bits[w] |= bit;
        return was;
    }
    return !b.rows_far.insert(row).second;
}
// pinned staging chunks are expensive to allocate and free (milliseconds each): one set is kept for the next fill-mode matrix
struct FillStagingCache { std::mutex mu; int64_t* hI[2] = {nullptr, nullptr}; int64_t* hJ[2] = {nullptr, nullptr}; double* hV[2] = {nullptr, nullptr}; int device = -1; };
static FillStagingCache g_fill_cache;
static void fill_release(FillBuffer& b) {
    if (b.stream) hipStreamSynchronize(b.stream);
    {
        std::lock_guard<std::mutex> lk(g_fill_cache.mu);
        if (b.hI[0] && g_fill_cache.hI[0] == nullptr) {
            for (int k = 0; k < 2; ++k) { g_fill_cache.hI[k] = b.hI[k]; g_fill_cache.hJ[k] = b.hJ[k]; g_fill_cache.hV[k] = b.hV[k]; b.hI[k] = nullptr; b.hJ[k] = nullptr; b.hV[k] = nullptr; }
            g_fill_cache.device = b.device;
        }
    }
    for (int k = 0; k < 2; ++k) {
        if (b.hI[k]) hipHostFree(b.hI[k]);
        if (b.hJ[k]) hipHostFree(b.hJ[k]);
        if (b.hV[k]) hipHostFree(b.hV[k]);
        if (b.uploaded[k]) hipEventDestroy(b.uploaded[k]);
    }
    pool_free(b.dI); pool_free(b.dJ); pool_free(b.dV);
    pool_free(b.d_acc); pinned_free(b.h_acc);
    if (b.stream) stream_put(b.stream, b.device);
    b = FillBuffer();
}
struct dsa_mat {
    int64_t m = 0, n = 0;
    bool fillmode = false;
    FillBuffer buf;
    bool has_major = false;
    Pma col, row;          // colmajor / rowmajor MappedPackedCSC
    double* d_x = nullptr; double* d_y = nullptr; int64_t x_cap = 0, y_cap = 0;
    // sparse-x product (sparsex.hip): acc / bm keep a ZERO INVARIANT between two products; everything grown, never shrunk
    struct Spx {
        double* acc = nullptr; uint64_t* bm = nullptr; int64_t rows_cap = 0;      // sums per row, one bit per touched row
        uint32_t* tile_cnt = nullptr; uint32_t* tile_off = nullptr; unsigned int* ticket = nullptr; int64_t tiles_cap = 0;
        int64_t* oi = nullptr; double* ov = nullptr; int64_t out_cap = 0;         // packed result in HBM
        int64_t* dx = nullptr; int64_t x_cap = 0;                                 // xi | xv uploaded
        int64_t* d_count = nullptr;
        long long* pin = nullptr;                                                 // landing area: 8 header words + 2 x SPX_PIN_CELLS
        void* stage = nullptr; size_t stage_bytes = 0;                            // pinned staging: x on the way up, long results on the way down
        unsigned long long seq = 0;
        std::vector<hipEvent_t> ev;                                               // behind the pieces of a long result on their way down
        bool dl_started = false; int dl_np = 0; size_t dl_off[8] = {}, dl_bytes[8] = {}; // ... its pieces: offset in [rows | values], bytes
        int64_t res_count = -1;                                                   // result of the last begin (-1: none)
        hipStream_t res_stream = nullptr;
    } spx;
    std::vector<int64_t> pi, pj; std::vector<double> pv;      // queued single writes (non-fill mode)
};

namespace {

// both orientations from (row, col, value) triples that are ALREADY in HBM (they stay the caller's): dynamicsparse(I, J, V) after its
// upload, closefillmode! straight from the device-resident fill buffer.  On failure nothing of the two structures is left behind.
void mat_build_major_dev(dsa_mat* h, const int64_t* dI, const int64_t* dJ, const double* dV, int64_t nnz, bool wide_rows, bool wide_cols,
                         KeyRange rows = KeyRange(), KeyRange cols = KeyRange()) {
    try {
        static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
        const auto ti0 = std::chrono::steady_clock::now();
        pma_init_common(h->col, true, true);
        pma_init_common(h->row, true, true);
        const auto tb0 = std::chrono::steady_clock::now();
        if (dbg_time) fprintf(stderr, "[mat_build_major] handles (streams, control blocks) %.2f ms\n", std::chrono::duration<double, std::milli>(tb0 - ti0).count());
        // the two orientations are independent structures on their own streams and both only read the triples: built side by side
        // (the rowmajor one on a helper thread; each build waits once for its cell / partition counts)
        // large builds: ONE sort of the triples, the rowmajor orientation from the cells colmajor's emit leaves behind (mat_build_both_dev).
        // DSA_BUILD_TWIN=0: two independent builds side by side, as in rounds 3-5 (A/B, coverage)
        static const bool twin = [] { const char* e = dev_env("DSA_BUILD_TWIN"); return !(e && e[0] == '0'); }();
        if (twin && nnz >= (1 << 16) && h->col.stream != h->row.stream && dev_env("DSA_FAIL_BUILD") == nullptr) {
            const bool both = mat_build_both_dev(h->col, h->row, dJ, dI, dV, nnz, wide_rows, wide_cols, cols, rows);
            if (!both) pma_build_dev(h->row, dI, dJ, dV, nnz, DSA_COMBINE_ADD, 0, 0, wide_cols, rows, cols);
            if (dbg_time) fprintf(stderr, "[mat_build_major] nnz=%lld both orientations (%s) %.1f ms\n", (long long)nnz, both ? "twin derived" : "general path",
                                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
            h->has_major = true;
            return;
        }
        std::exception_ptr row_exc;
        std::thread row_thread;
        const bool side_by_side = h->col.stream != h->row.stream && nnz > 0;
        if (side_by_side)
            row_thread = std::thread([&] {
                try { bind_device(h->row); pma_build_dev(h->row, dI, dJ, dV, nnz, DSA_COMBINE_ADD, 0, 0, wide_cols, rows, cols); }
                catch (...) { row_exc = std::current_exception(); }
            });
        try { pma_build_dev(h->col, dJ, dI, dV, nnz, DSA_COMBINE_ADD, 0, 0, wide_rows, cols, rows); }      // dynamicsparsecolmajor(I, J, V): partitions = columns, keys = rows
        catch (...) { if (row_thread.joinable()) row_thread.join(); throw; }
        const auto tb1 = std::chrono::steady_clock::now();
        if (side_by_side) { row_thread.join(); if (row_exc) std::rethrow_exception(row_exc); }
        else pma_build_dev(h->row, dI, dJ, dV, nnz, DSA_COMBINE_ADD, 0, 0, wide_cols, rows, cols);      // dynamicsparsecolmajor(J, I, V): partitions = rows, keys = columns
        if (dbg_time)
            fprintf(stderr, "[mat_build_major] nnz=%lld colmajor %.1f ms  rowmajor %.1f ms\n", (long long)nnz,
                    std::chrono::duration<double, std::milli>(tb1 - tb0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb1).count());
    } catch (...) {
        // streams, control blocks and whatever slot buffers / tables exist by now (dsa_mat_destroy only looks at them once has_major is set)
        pma_destroy(h->col);
        pma_destroy(h->row);
        throw;
    }
    h->has_major = true;
}

void check_key(int64_t k) { if (k == 0) fail(DSA_EKEY, "0 is the reserved semaphore key (src/pcsr.jl:23)"); }

// dynamicsparse(I, J, V): upload + both builds; *rows_out / *cols_out receive the value ranges of the keys
void mat_build_major(dsa_mat* h, const int64_t* I, const int64_t* J, const double* V, int64_t nnz, KeyRange* rows_out = nullptr, KeyRange* cols_out = nullptr) {
    int64_t *dI = nullptr, *dJ = nullptr; double* dV = nullptr;
    const auto tup0 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(g_device));
    try {
        KeyRange rows, cols;
        bool zr = false, zc = false;
        if (nnz > 0) {
            HIPCHK(pool_alloc(reinterpret_cast<void**>(&dI), (size_t)nnz * sizeof(int64_t)));
            HIPCHK(pool_alloc(reinterpret_cast<void**>(&dJ), (size_t)nnz * sizeof(int64_t)));
            HIPCHK(pool_alloc(reinterpret_cast<void**>(&dV), (size_t)nnz * sizeof(double)));
            HIPCHK(hipMemcpy(dI, I, (size_t)nnz * sizeof(int64_t), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(dJ, J, (size_t)nnz * sizeof(int64_t), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(dV, V, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
            // value ranges (K-build's composite, the storage width of the keys, the size guesses) and the reserved key 0: one pass on
            // the device over the arrays just uploaded — the host never loops over the caller's 10^7 keys
            hipError_t e = device_key_scan(dI, dJ, nnz, &rows, &cols, &zr, &zc, nullptr);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("key scan: ") + hipGetErrorString(e));
        }
        static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
        if (dbg_time) fprintf(stderr, "[mat_build_major] upload of %lld triples from caller memory + key scan %.1f ms\n", (long long)nnz,
                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tup0).count());
        if (rows_out) *rows_out = rows;
        if (cols_out) *cols_out = cols;
        if (zr || zc) check_key(0);
        auto fit32 = [](const KeyRange& r) { return !g_force_wide && (!r.known() || (key_fits32(r.lo) && key_fits32(r.hi))); };
        mat_build_major_dev(h, dI, dJ, dV, nnz, !fit32(rows), !fit32(cols), rows, cols);
    } catch (...) { pool_free(dI); pool_free(dJ); pool_free(dV); throw; }      // (a failed build has synchronised and destroyed its streams)
    pool_free(dI); pool_free(dJ); pool_free(dV);
}

Pma& orient(dsa_mat* h, int32_t o) {
    if (!h->has_major) fail(DSA_EMODE, "matrix is in fill mode");
    if (o != DSA_COLMAJOR && o != DSA_ROWMAJOR) fail(DSA_EARG, "orientation must be 0 or 1");
    return o == DSA_COLMAJOR ? h->col : h->row;
}

Op make_op(int32_t kind, int64_t a, int64_t b, double v) { Op o; o.a = a; o.b = b; o.v = v; o.kind = kind; o.pad = 0; return o; }

// Host replay of ONE partition table (MappedPackedCSC.col_keys + which ids are tombstones) through a batch of writes: which write is
// the first one the reference refuses?  Control logic like the integer density bounds — no slot is read or written here; the device
// executes (and judges) every write.  Follows find(col_keys, col) + addcolumn! + the table half of addpartition!(pcsc, prev)
// (src/pcsr.jl:341-351, 148-169, 114-146): a key that is present changes nothing; a key behind the last entry is pushed; a key whose
// table slot prev+1 is a tombstone reuses it, unless no live id follows (semaphores[0]: BoundsError, :121-125); a key in front of an
// occupied slot shifts the tail of the tables, which asserts on the first tombstone it meets (:129-133).
struct TableReplay {
    std::vector<int64_t> ck; std::vector<uint8_t> live;
    int64_t last_tomb = 0, last_live = 0;          // 1-based index of the last tombstone / the last live entry (0: none)
    void load(Pma& P) {
        const int64_t len = P.h_ctl->table_len;
        ck.resize((size_t)len); live.resize((size_t)len);
        if (len > 0) {
            HIPCHK(hipMemcpyAsync(ck.data(), P.col_keys, (size_t)len * sizeof(int64_t), hipMemcpyDeviceToHost, P.stream));
            HIPCHK(hipMemcpyAsync(live.data(), P.col_live, (size_t)len, hipMemcpyDeviceToHost, P.stream));
            HIPCHK(hipStreamSynchronize(P.stream));
        }
        rescan();
    }
    void rescan() {
        last_tomb = 0; last_live = 0;
        for (int64_t i = (int64_t)live.size(); i >= 1 && (last_tomb == 0 || last_live == 0); --i) {
            if (live[(size_t)i - 1]) { if (last_live == 0) last_live = i; } else if (last_tomb == 0) last_tomb = i;
        }
    }
    // find(col_keys, key) with tombstones (the host twin of d_find_table, csrc/find_dev.h): position of the key or of its predecessor
    int64_t find(int64_t key, bool* exact) const {
        int64_t from = 1, to = (int64_t)ck.size();
        *exact = false;
        while (from <= to) {
            const int64_t mid = (from + to) >> 1;
            int64_t i = mid;
            while (i >= from && !live[(size_t)i - 1]) --i;
            if (i < from) from = mid + 1;
            else {
                const int64_t c = ck[(size_t)i - 1];
                if (c > key) to = i - 1;
                else if (c < key) from = mid + 1;
                else { *exact = true; return i; }
            }
        }
        int64_t i = to;
        while (i > 0 && !live[(size_t)i - 1]) --i;
        return i;
    }
    // a write into partition `key`: 0 when the table accepts it (and the table as it is afterwards), else the reference's error
    int32_t touch(int64_t key) {
        bool exact = false;
        const int64_t prev = find(key, &exact);
        if (exact) return 0;
        const int64_t len = (int64_t)ck.size();
        if (prev == len) { ck.push_back(key); live.push_back(1); last_live = len + 1; return 0; }
        if (!live[(size_t)prev]) {                                  // entry prev + 1 is a tombstone: reuse it
            if (last_live <= prev + 1) return DSA_EBOUNDS;
            ck[(size_t)prev] = key; live[(size_t)prev] = 1;
            if (last_tomb == prev + 1) rescan();
            return 0;
        }
        if (last_tomb >= prev + 1) return DSA_EASSERT;               // the shift of the tail meets a tombstone
        ck.insert(ck.begin() + prev, key); live.insert(live.begin() + prev, (uint8_t)1);
        last_live = len + 1;
        return 0;
    }
};

// setindex! on both orientations for ops [0, n)  src/matrix.jl:53-59
void mat_apply_sets(dsa_mat* h, const int64_t* I, const int64_t* J, const double* V, int64_t n) {
    // colmajor[row, col] = val / rowmajor[col, row] = val: the batch-parallel path takes the caller's columns as they are, the other
    // branches below build the op arrays they need
    const OpBatch bc(OP_MPCSC_SET, I, J, V, n), br(OP_MPCSC_SET, J, I, V, n);
    std::vector<Op> oc, orw;
    auto build_ops = [&] {
        if (!oc.empty() || n == 0) return;
        oc.resize((size_t)n); orw.resize((size_t)n);
        for (int64_t k = 0; k < n; ++k) {
            oc[(size_t)k] = make_op(OP_MPCSC_SET, I[k], J[k], V[k]);
            orw[(size_t)k] = make_op(OP_MPCSC_SET, J[k], I[k], V[k]);
        }
    };
    // Without tombstones an OP_MPCSC_SET cannot fail (the reference's assert / bounds paths of addpartition! need a
    // deleted partition, App. A.6 (3)), so the two orientations can be updated concurrently; otherwise the colmajor
    // batch runs first and the rowmajor batch is cut at the failing op, like the reference's statement order.
    const bool no_tombstones = h->col.h_ctl->nb_partitions == h->col.h_ctl->table_len &&
                               h->row.h_ctl->nb_partitions == h->row.h_ctl->table_len;
    static const bool par = [] { const char* e = dev_env("DSA_PARBATCH"); return !(e && e[0] == '0'); }();
    // With tombstones a write can only fail while it CREATES a partition in the middle of the table (addpartition!(pcsc, prev),
    // src/pcsr.jl:114-146: the two asserts, and the lookup behind a tombstoned tail).  A batch whose partition keys never decrease and start
    // at or behind the last table entry — which must be live — only ever writes to that entry or appends behind it (find() returns the
    // last position, addcolumn! takes its push! branch, src/pcsr.jl:148-154): it cannot fail either, and the two orientations may run side
    // by side.  That is column generation with deletions: new columns get new, larger ids while old ones are deleted (config 5 + deletecolumn!:
    // 445 -> 330 ms).  Anything else with tombstones keeps the reference's statement order below.
    auto cannot_fail = [&](Pma& P, const int64_t* part_keys_of_ops) -> bool {
        const Ctl& c = *P.h_ctl;
        if (c.nb_partitions == c.table_len) return true;
        if (c.table_len <= 0 || c.n_pending != 0) return false;
        uint8_t live = 0; int64_t last_key = 0;
        HIPCHK(hipMemcpyAsync(&live, P.col_live + (c.table_len - 1), 1, hipMemcpyDeviceToHost, P.stream));
        HIPCHK(hipMemcpyAsync(&last_key, P.col_keys + (c.table_len - 1), sizeof(int64_t), hipMemcpyDeviceToHost, P.stream));
        HIPCHK(hipStreamSynchronize(P.stream));
        if (!live) return false;
        int64_t running = last_key;
        for (int64_t k = 0; k < n; ++k) { if (part_keys_of_ops[k] < running) return false; running = part_keys_of_ops[k]; }
        return true;
    };
    static const bool tomb_par = [] { const char* e = dev_env("DSA_TOMBSTONE_PAR"); return !(e && e[0] == '0'); }();
    const bool side_by_side_ok = no_tombstones || (tomb_par && n >= 128 && cannot_fail(h->col, J) && cannot_fail(h->row, I));
    if (par && side_by_side_ok && n >= 128) {
        // batch-parallel rounds per orientation (writes to existing columns with disjoint footprints run concurrently; new
        // columns and anything else fall back to the sequential sequencer inside run_ops_parallel)
        int32_t ec = 0, er = 0;
        static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
        // the two orientations are independent structures on their own streams: their round / sequencer loops (host-driven)
        // run side by side, the rowmajor one on a helper thread
        const auto t0 = std::chrono::steady_clock::now();
        int64_t dc = 0, dr = 0;
        std::exception_ptr row_exc;
        auto t_row_done = t0;
        const bool side_by_side = h->col.stream != h->row.stream;
        std::thread row_thread;
        if (side_by_side)
            row_thread = std::thread([&] {
                try { bind_device(h->row); dr = run_ops_parallel(h->row, br, &er); t_row_done = std::chrono::steady_clock::now(); }
                catch (...) { row_exc = std::current_exception(); }
            });
        const int64_t rounds0 = h->row.stat_par_rounds;
        try { dc = run_ops_parallel(h->col, bc, &ec); }
        catch (...) { if (row_thread.joinable()) row_thread.join(); throw; }
        const auto t1 = std::chrono::steady_clock::now();
        if (side_by_side) { row_thread.join(); if (row_exc) std::rethrow_exception(row_exc); }
        else dr = run_ops_parallel(h->row, br, &er);
        if (dbg_time)
            fprintf(stderr, "[mat_apply_sets] n=%lld both orientations %.2f ms (colmajor done after %.2f ms, rowmajor after %.2f ms)  colmajor (par %lld seq %lld ext %lld)  "
                    "rowmajor (par %lld seq %lld ext %lld rounds +%lld)\n", (long long)n,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t_row_done - t0).count(), (long long)h->col.stat_par_ops,
                    (long long)h->col.stat_seq_ops, (long long)h->col.h_ctl->stat_extends, (long long)h->row.stat_par_ops, (long long)h->row.stat_seq_ops,
                    (long long)h->row.h_ctl->stat_extends, (long long)(h->row.stat_par_rounds - rounds0));
        if (dbg_time)
            fprintf(stderr, "    rowmajor: sequencer launches %lld  stops by reason [0 unplannable %lld, 1 newcol %lld, 2 limits %lld, 3 shifts %lld, 4 semleaf %lld, "
                    "5 window %lld, 6 scan %lld, 7 conflict %lld]\n", (long long)h->row.stat_seq_launches, (long long)h->row.stat_why[0], (long long)h->row.stat_why[1],
                    (long long)h->row.stat_why[2], (long long)h->row.stat_why[3], (long long)h->row.stat_why[4], (long long)h->row.stat_why[5],
                    (long long)h->row.stat_why[6], (long long)h->row.stat_why[7]);
        if (dbg_time && h->row.h_ctl->prof[7] != 0) {      // -DDSA_PROFILE builds only: shader-clock split of the rowmajor sequencer (cumulative)
            const int64_t* q = h->row.h_ctl->prof;
            const double us = 1.0 / 100.0;                   // s_memtime ticks at 100 MHz
            fprintf(stderr, "    rowmajor sequencer profile (cumulative): kernel %.0f us | %lld matrix writes: table lookup %.0f us, %lld new partitions %.0f us, write in partition %.0f us "
                    "[partition end %.0f, find %.0f, insert/shift %.0f, scan+rebalance %.0f (%lld in-block rebalances %.0f us, %lld slots)] | %lld table merges %.0f us\n",
                    q[7] * us, (long long)q[4], q[0] * us, (long long)q[5], q[1] * us, q[2] * us, q[11] * us, q[8] * us, q[9] * us, q[10] * us, (long long)q[13], q[12] * us,
                    (long long)q[14], (long long)q[6], q[3] * us);
        }
        const int64_t done = std::min(dc, dr);
        for (int64_t k = 0; k < std::min(done + 1, n); ++k) if (V[k] != 0.0) { h->m = std::max(h->m, I[k]); h->n = std::max(h->n, J[k]); }
        if (ec && er && dr < dc) fail(er, err_text(er));      // both halves failed: the rowmajor half of the EARLIER write comes first
        if (ec) fail(ec, err_text(ec));
        if (er) fail(er, err_text(er));
        return;
    }
    build_ops();
    if (no_tombstones && h->col.stream != h->row.stream) {
        // A small batch (a column or a few that arrive together, a row): the orientation in which its writes fall into MANY
        // partitions takes the local rounds (one wave per op: k_local_rounds), the one in which they share a few partitions — writes
        // into one column are ordered by nature — its sequencer, side by side.  16 writes of a new column: 218 -> ~120 us.
        static const bool small_rounds = [] { const char* e = dev_env("DSA_SMALL_ROUNDS"); return !(e && e[0] == '0'); }();
        if (small_rounds && par && n >= 8) {
            std::vector<int64_t> di(I, I + n), dj(J, J + n);
            std::sort(di.begin(), di.end()); std::sort(dj.begin(), dj.end());
            const int64_t ni = std::unique(di.begin(), di.end()) - di.begin(), nj = std::unique(dj.begin(), dj.end()) - dj.begin();
            Pma* rounds = nullptr; Pma* seq = nullptr; const std::vector<Op>* ro = nullptr; const std::vector<Op>* so = nullptr;
            if (ni >= 2 * nj && ni >= 8) { rounds = &h->row; ro = &orw; seq = &h->col; so = &oc; }          // many rows, few columns
            else if (nj >= 2 * ni && nj >= 8) { rounds = &h->col; ro = &oc; seq = &h->row; so = &orw; }     // many columns, few rows
            if (rounds != nullptr) {
                SeqRun rs;
                seq_start(rs, *seq, *so);
                int32_t er = 0;
                int64_t dr = 0;
                try { dr = run_ops_parallel(*rounds, *ro, &er); } catch (...) { while (seq_step(rs)) {} throw; }
                while (seq_step(rs)) {}
                const int64_t done = std::min(dr, rs.applied);
                for (int64_t k = 0; k < std::min(done + 1, n); ++k) if (V[k] != 0.0) { h->m = std::max(h->m, I[k]); h->n = std::max(h->n, J[k]); }
                if (rs.err) fail(rs.err, err_text(rs.err));
                if (er) fail(er, err_text(er));
                return;
            }
        }
        SeqRun rc, rr;
        run_ops_pair(h->col, oc, h->row, orw, rc, rr);
        const int64_t done = std::min(rc.applied, rr.applied);
        for (int64_t k = 0; k < std::min(done + 1, n); ++k) if (V[k] != 0.0) { h->m = std::max(h->m, I[k]); h->n = std::max(h->n, J[k]); }
        if (rc.err) fail(rc.err, err_text(rc.err));
        if (rr.err) fail(rr.err, err_text(rr.err));
        return;
    }
    // tombstones present: a write that creates a column can fail (src/pcsr.jl:124,132), so the colmajor batch runs first and the
    // rowmajor batch is cut at the failing op, like the reference's statement order — still through the batch-parallel
    // rounds for everything that is plannable (writes to existing columns); new columns take the sequencer's literal path.
    //
    // The state after a FAILED batch is the reference's too (round 6; until then colmajor could hold writes behind the failing one):
    // the reference stops at write k with colmajor holding writes [0, k] when its rowmajor statement threw, [0, k) when the colmajor one
    // did, rowmajor [0, k) either way (src/matrix.jl:43-62).  Colmajor running ahead of rowmajor is only wrong when ROWMAJOR fails, and
    // whether a write fails depends on nothing but the partition table (which ids are tombstones: src/pcsr.jl:114-162).  So when the
    // rowmajor table holds tombstones the first write it will refuse is found up front by replaying the table — not the slots — on the
    // host (TableReplay), and colmajor is not run beyond it.  The device stays the judge: the error code comes from the device, and
    // when the device accepts what the replay predicted to fail the loop simply goes on behind it.
    const bool par_seq = par && n >= 128;
    int64_t pos = 0;
    while (pos < n) {
        const int64_t rem = n - pos;
        int64_t f = n;                                                  // first write the rowmajor table refuses (n: none)
        const Ctl& rc = *h->row.h_ctl;
        if (rem > 1 && rc.nb_partitions != rc.table_len) {
            TableReplay tr;
            tr.load(h->row);
            for (int64_t k = pos; k < n; ++k) if (tr.touch(I[k]) != 0) { f = k; break; }
        }
        const int64_t end = std::min(n, f + 1);
        std::vector<Op> pc(oc.begin() + pos, oc.begin() + end), pr(orw.begin() + pos, orw.begin() + end);
        int32_t err = 0;
        const int64_t done = (par_seq ? run_ops_parallel(h->col, pc, &err, true) : run_ops(h->col, pc, &err)) + pos;
        if (err) {
            // colmajor refused write `done` (<= f): rowmajor gets the writes in front of it — none of which its table refuses
            pr.resize((size_t)(done - pos));
            int32_t e2 = 0;
            const int64_t d2 = (par_seq ? run_ops_parallel(h->row, pr, &e2, true) : run_ops(h->row, pr, &e2)) + pos;
            // Which write fails FIRST in the reference's order (colmajor then rowmajor of write 0, of write 1, ...): the rowmajor half of an
            // earlier write d2 < done comes before the colmajor half of write `done`.  (Rounds 1-4 reported the colmajor error regardless:
            // tools/fuzz.py run_tombstones seed 503707, EBOUNDS where the reference throws the AssertionError of src/pcsr.jl:132 three writes earlier.)
            const int64_t fail_at = e2 ? d2 : done;
            // the failing write had already updated size(m) in the reference (src/matrix.jl:44-47)
            for (int64_t k = 0; k <= fail_at && k < n; ++k) if (V[k] != 0.0) { h->m = std::max(h->m, I[k]); h->n = std::max(h->n, J[k]); }
            if (e2) fail(e2, err_text(e2));
            fail(err, err_text(err));
        }
        int32_t e2 = 0;
        const int64_t d2 = (par_seq ? run_ops_parallel(h->row, pr, &e2, true) : run_ops(h->row, pr, &e2)) + pos;
        if (e2) {
            for (int64_t k = 0; k <= d2 && k < n; ++k) if (V[k] != 0.0) { h->m = std::max(h->m, I[k]); h->n = std::max(h->n, J[k]); }
            fail(e2, err_text(e2));
        }
        for (int64_t k = pos; k < end; ++k) if (V[k] != 0.0) { h->m = std::max(h->m, I[k]); h->n = std::max(h->n, J[k]); }
        pos = end;
    }
}

// range_from > 0: the stored cells of the slot range [range_from, range_to] instead of the column `col` (at most VIEW_SMALL_SLOTS slots)
void view_small(Pma& P, int64_t col, int64_t range_from, int64_t range_to, std::vector<int64_t>& ks, std::vector<double>& vs) {
    // one launch (partition lookup + K-pack of its slot range into the idle alternate buffer) and one host round trip for
    // partitions of up to VIEW_SMALL_SLOTS slots; the first SPEC cells travel with the meta words, longer views fetch the rest
    constexpr int64_t SPEC = 512;
    ks.clear(); vs.clear();
    const int alt = 1 - P.cur;
    const int64_t out_cap = std::min<int64_t>(P.cap_alloc, VIEW_SMALL_SLOTS);
    const int64_t spec = std::min<int64_t>(SPEC, out_cap);
    int64_t r[5] = {0, 0, 0, 0, 0};
    if (publish_enabled()) {
        // the kernel writes the meta words and the first SPEC cells straight into a pinned landing area and then a sequence number: the host
        // polls for it — no copy command, no stream synchronisation (60 -> 20 us per view; DSA_PUBLISH=0 = copies + synchronisation)
        ViewAreaLease lease(P);
        const unsigned long long seq = ++P.view_seq;
        hipError_t e = launch_view_small(P.K(), P.V(), P.O(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len, P.capacity(), col,
                                         P.KA(alt), P.vals[alt], out_cap, P.d_small, P.h_view, SPEC, seq, range_from, range_to, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("view launch: ") + hipGetErrorString(e));
        wait_policy_block(P);
        volatile int64_t* seqp = P.h_view + 5;
        auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        while ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) {
            if (std::chrono::steady_clock::now() < next_query) continue;
            const hipError_t q = hipStreamQuery(P.stream);
            if (q == hipErrorNotReady) { next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2); continue; }
            if (q != hipSuccess) fail(DSA_EHIP, std::string("view: ") + hipGetErrorString(q));
            if ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) fail(DSA_EHIP, "view kernel finished without publishing its result");
        }
        for (int q = 0; q < 5; ++q) r[q] = P.h_view[q];
        const int64_t have = std::max<int64_t>(0, std::min<int64_t>(r[4], spec));
        ks.assign(P.h_view + 8, P.h_view + 8 + have);
        vs.resize((size_t)have);
        std::memcpy(vs.data(), P.h_view + 8 + SPEC, (size_t)have * sizeof(double));
        ks.resize((size_t)spec); vs.resize((size_t)spec);
    } else {
        hipError_t e = launch_view_small(P.K(), P.V(), P.O(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len, P.capacity(), col,
                                         P.KA(alt), P.vals[alt], out_cap, P.d_small, nullptr, 0, 0ull, range_from, range_to, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("view launch: ") + hipGetErrorString(e));
        ks.resize((size_t)spec); vs.resize((size_t)spec);
        HIPCHK(hipMemcpyAsync(r, P.d_small, sizeof(r), hipMemcpyDeviceToHost, P.stream));
        HIPCHK(hipMemcpyAsync(vs.data(), P.vals[alt], (size_t)spec * sizeof(double), hipMemcpyDeviceToHost, P.stream));
        download_keys(P, ks.data(), P.keys[alt], spec);       // synchronises
    }
    if (r[2] != 0) { ks.clear(); vs.clear(); fail((int32_t)r[2], "partition has no semaphore"); }
    if (r[0] == 0) { ks.clear(); vs.clear(); return; }       // empty view: the column does not exist (src/views.jl:17,24)
    const int64_t cnt = r[4];
    if (cnt < 0) { read_range_general(P, r[0], r[1], ks, vs); return; }   // a long partition: general K-pack path
    ks.resize((size_t)cnt); vs.resize((size_t)cnt);
    if (cnt > spec) {
        HIPCHK(hipMemcpyAsync(vs.data() + spec, P.vals[alt] + spec, (size_t)(cnt - spec) * sizeof(double), hipMemcpyDeviceToHost, P.stream));
        download_keys(P, ks.data() + spec, (char*)P.keys[alt] + (size_t)spec * P.kb(), cnt - spec);
    }
}

void col_view_of(Pma& P, int64_t col, std::vector<int64_t>& ks, std::vector<double>& vs) { view_small(P, col, 0, 0, ks, vs); }

// view(mpcsc, :, col) (src/views.jl:15-35) that stays in HBM: the stored cells of the column packed, in slot order, at the front of
// P's idle alternate buffer (P.KA(1 - P.cur), P.vals[1 - P.cur]); only the meta words reach the host (through the pinned landing area:
// no copy command).  cnt = number of cells, last_key = key of the last one (the largest: a partition is key-ordered).
struct DevView { int64_t cnt = 0, last_key = 0; };
DevView view_dev(Pma& P, int64_t col) {
    DevView dv;
    const int alt = 1 - P.cur;
    const int64_t out_cap = std::min<int64_t>(P.cap_alloc, VIEW_SMALL_SLOTS);
    int64_t r[6] = {0, 0, 0, 0, 0, 0};
    if (publish_enabled()) {
        ViewAreaLease lease(P);
        const unsigned long long seq = ++P.view_seq;
        hipError_t e = launch_view_small(P.K(), P.V(), P.O(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len, P.capacity(), col,
                                         P.KA(alt), P.vals[alt], out_cap, P.d_small, P.h_view, 0, seq, 0, 0, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("view launch: ") + hipGetErrorString(e));
        wait_view_seq(P, seq, "view");
        for (int q = 0; q < 5; ++q) r[q] = P.h_view[q];
        r[5] = P.h_view[6];
    } else {
        hipError_t e = launch_view_small(P.K(), P.V(), P.O(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len, P.capacity(), col,
                                         P.KA(alt), P.vals[alt], out_cap, P.d_small, nullptr, 0, 0ull, 0, 0, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("view launch: ") + hipGetErrorString(e));
        HIPCHK(hipMemcpyAsync(P.h_small, P.d_small, sizeof(r), hipMemcpyDeviceToHost, P.stream));
        HIPCHK(hipStreamSynchronize(P.stream));
        for (int q = 0; q < 6; ++q) r[q] = P.h_small[q];
    }
    if (r[2] != 0) fail((int32_t)r[2], "partition has no semaphore");
    if (r[0] == 0) return dv;                                  // the column does not exist (src/views.jl:17,24)
    dv.cnt = r[4]; dv.last_key = r[5];
    if (dv.cnt < 0) {                                          // a long partition: tile counts + scan + K-pack; the count and one key come back
        int64_t cnt = 0;
        hipError_t e = launch_compact_range(P.K(), P.V(), P.O(), r[0], r[1], P.KA(alt), P.vals[alt], P.cap_alloc, &P.work, &cnt, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("compact launch: ") + hipGetErrorString(e));
        dv.cnt = cnt; dv.last_key = 0;
        if (cnt > 0) {
            HIPCHK(hipMemcpyAsync(P.h_small, (const char*)P.keys[alt] + (size_t)(cnt - 1) * P.kb(), P.kb(), hipMemcpyDeviceToHost, P.stream));
            HIPCHK(hipStreamSynchronize(P.stream));
            dv.last_key = P.wide ? P.h_small[0] : (int64_t) * reinterpret_cast<const int32_t*>(P.h_small);
        }
    }
    return dv;
}

void ensure_xy(dsa_mat* h, int64_t nx, int64_t ny) {
    if (nx > h->x_cap) { if (h->d_x) hipFree(h->d_x); h->x_cap = std::max<int64_t>(nx, 1024); HIPCHK(hipMalloc(&h->d_x, (size_t)h->x_cap * sizeof(double))); }
    if (ny > h->y_cap) { if (h->d_y) hipFree(h->d_y); h->y_cap = std::max<int64_t>(ny, 1024); HIPCHK(hipMalloc(&h->d_y, (size_t)h->y_cap * sizeof(double))); }
}

// mat * v walks the colmajor orientation in the reference (src/operations.jl:14-24), transpose(mat) * v the
// rowmajor one (:26-36).  Gather form: the twin orientation, whose partitions are the OUTPUT index.
// What the gather launch may assume about an orientation (recomputed after every launch that can change the layout or
// the tables: one small kernel + an 8-byte round trip, amortised over the SpMV calls between two write batches).
// Tables with tombstones or unmerged entries take the memset path whatever the layout: nothing to compute.
bool spmv_meta_applicable(const Pma& P) {
    const Ctl& c = *P.h_ctl;
    return P.has_cols && c.table_len > 0 && c.nb_partitions == c.table_len && c.n_pending == 0;
}
// Enqueues k_spmv_meta + the copy of its 5 result words behind whatever is on the stream (one launch, no host wait).  Called at the
// end of every write batch / build, so that the product that follows finds the words already in pinned memory: the product after a
// write batch costs what its kernel costs (round 2: a 359 us meta kernel + a host round trip in front of an 8.7 us SpMV in config 5).
void prefetch_spmv_meta(Pma& P) {
    if (P.spmv_meta.epoch == P.layout_epoch || P.meta_inflight_epoch == P.layout_epoch || !spmv_meta_applicable(P)) return;
    if (!P.d_meta) {
        HIPCHK(hipMalloc(&P.d_meta, SPMV_META_WORDS * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(P.d_meta, 0, SPMV_META_WORDS * sizeof(unsigned long long), P.stream));
        HIPCHK(pinned_alloc(reinterpret_cast<void**>(&P.h_meta), 8 * sizeof(int64_t)));
        std::memset(P.h_meta, 0, 8 * sizeof(int64_t));
        P.meta_seq = 0;
    }
    { static const char* dbg = dev_env("DSA_DBG_SPMV_META"); if (dbg) fprintf(stderr, "prefetch_spmv_meta: launch for epoch %lld (cached %lld, in flight %lld)\n",
                                                                               (long long)P.layout_epoch, (long long)P.spmv_meta.epoch, (long long)P.meta_inflight_epoch); }
    // the kernel writes its five words and then the sequence number straight into pinned host memory
    hipError_t e = launch_spmv_meta(P.sems, P.col_keys, P.h_ctl->table_len, P.h_ctl->capacity, P.d_meta,
                                    reinterpret_cast<unsigned long long*>(P.h_meta), ++P.meta_seq, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("spmv meta launch: ") + hipGetErrorString(e));
    P.meta_inflight_epoch = P.layout_epoch;
}
const Pma::SpmvMeta& spmv_meta(Pma& P) {
    Pma::SpmvMeta& M = P.spmv_meta;
    if (M.epoch == P.layout_epoch) return M;
    M = Pma::SpmvMeta();
    M.epoch = P.layout_epoch;
    static const char* dbg = dev_env("DSA_DBG_SPMV_META");
    if (!spmv_meta_applicable(P)) {     // tombstones: memset path
        const Ctl& c = *P.h_ctl;
        if (dbg) fprintf(stderr, "spmv_meta: has_cols %d table_len %lld nb_partitions %lld n_pending %lld\n", (int)P.has_cols, (long long)c.table_len,
                         (long long)c.nb_partitions, (long long)c.n_pending);
        return M;
    }
    M.epoch = -1;
    prefetch_spmv_meta(P);                      // no-op when the write batch has already enqueued it
    // wait for the sequence number: normally there already (the kernel was enqueued behind the write batch); a stream wait if it
    // does not show up within a millisecond
    wait_policy_block(P);
    volatile int64_t* seqp = P.h_meta + 5;
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    while ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != P.meta_seq) {
        if (!synced && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) { HIPCHK(hipStreamSynchronize(P.stream)); synced = true; continue; }
        if (synced) fail(DSA_EHIP, "SpMV meta kernel finished without publishing its result");
    }
    M.epoch = P.layout_epoch;
    const int64_t* r = P.h_meta;
    M.ordered = r[4] == 0;
    M.max_extent = r[0]; M.max_gap = r[1]; M.first_key = r[2]; M.last_key = r[3];
    if (dbg) fprintf(stderr, "spmv_meta: table_len %lld ordered %d max_extent %lld max_gap %lld first %lld last %lld\n", (long long)P.h_ctl->table_len,
                     (int)M.ordered, (long long)M.max_extent, (long long)M.max_gap, (long long)M.first_key, (long long)M.last_key);
    return M;
}

void mat_prefetch_spmv_meta(dsa_mat* h) {
    if (!h->has_major) return;
    prefetch_spmv_meta(h->row);
    prefetch_spmv_meta(h->col);
}

void spmv_dev(dsa_mat* h, int32_t transpose, int32_t algo, const double* d_x, int64_t nx, double* d_y, int64_t ny, hipStream_t s,
              int pattern = 0) {
    if (!h->has_major) fail(DSA_EMODE, "matrix is in fill mode");
    hipError_t e;
    if (algo == 0) {
        Pma& P = transpose ? h->col : h->row;
        int mode = 0;
        // no memset of y when every row is written exactly once by the kernel: no partition longer than a span (no atomics)
        // and the rows without a partition are few (the owner of the next partition zeroes them)
        constexpr int64_t MAX_FILL = 4096;
        const Pma::SpmvMeta& M = spmv_meta(P);
        if (M.ordered && M.max_extent <= SPMV_SPAN_SLOTS && M.max_gap <= MAX_FILL && M.first_key >= 1 && M.first_key <= MAX_FILL &&
            ny - M.last_key <= MAX_FILL)
            { mode |= SPMV_ZFILL; ++P.stat_spmv_nomemset; }
        if (nx * (int64_t)sizeof(double) <= (3 << 20)) mode |= SPMV_PLAIN_STREAM;      // x stays in an XCD's 4 MB L2 beside the stream
        e = launch_spmv_gather(P.K(), P.V(), P.O(), P.capacity(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len, d_x, nx, d_y, ny, pattern, mode, s);
    } else if (algo == 1) {
        Pma& P = transpose ? h->row : h->col;
        e = launch_spmv_scatter(P.K(), P.V(), P.O(), P.capacity(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len, d_x, nx, d_y, ny, s);
    } else {
        fail(DSA_EARG, "algo must be 0 (gather) or 1 (scatter)");
    }
    if (e != hipSuccess) fail(DSA_EHIP, std::string("spmv launch: ") + hipGetErrorString(e));
}

}  // namespace

namespace dsa { void set_last_error(const char* msg) { g_err = msg ? msg : ""; } }

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* dsa_last_error_message(void) { return g_err.c_str(); }

int32_t dsa_device_count(int32_t* count) {
    API_TRY
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; fail(DSA_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    API_CATCH
}
// the caching allocator (pool.hip) keeps up to DSA_POOL_MAX_MB of idle HBM per process: a host that shares the card with another
// allocator (PyTorch's) asks how much that is and hands it back
int32_t dsa_pool_idle_bytes(int64_t* bytes) { API_TRY *bytes = (int64_t)pool_idle_bytes(); API_CATCH }
int32_t dsa_pool_trim(int64_t keep_bytes) { API_TRY pool_trim(keep_bytes > 0 ? (size_t)keep_bytes : 0); API_CATCH }
int32_t dsa_dev_switches(char* buf, int64_t cap, int32_t* enabled) {
    API_TRY
    std::string all;
    for (const char* s : dsa::k_dev_switches) { if (!all.empty()) all += ' '; all += s; }
    if (buf == nullptr || cap < (int64_t)all.size() + 1) fail(DSA_ECAP, "buffer too small for the list of development switches");
    std::memcpy(buf, all.c_str(), all.size() + 1);
    if (enabled) {
#ifdef DSA_DEV
        *enabled = 1;
#else
        const char* on = getenv("DSA_DEV");
        *enabled = (on && on[0] == '1') ? 1 : 0;
#endif
    }
    API_CATCH
}
int32_t dsa_set_device(int32_t device) {
    API_TRY
    HIPCHK(hipSetDevice(device));
    g_device = device;
    API_CATCH
}

// ---------------- vector ----------------
int32_t dsa_vec_create(const int64_t* keys, const double* vals, int64_t n, int32_t combine_op, int64_t len, dsa_vec_t** out) {
    API_TRY
    if (n < 0) fail(DSA_EARG, "negative length");
    if (len < 0) { len = 0; for (int64_t i = 0; i < n; ++i) len = std::max(len, keys[i]); }      // _guess_length  src/vector.jl:6
    if (n > 0xffffffffll) fail(DSA_EARG, "more than 2^32-1 entries in one call");
    auto* h = new dsa_vec();
    try {
        pma_init_common(h->P, false, false);
        // _prepare_keys_vals! (stable sort + left fold of duplicates, src/vector.jl:10-36) and the spread, on the device
        pma_build_from_host(h->P, nullptr, keys, vals, n, combine_op, 1, 0);
    } catch (...) { pma_destroy(h->P); delete h; throw; }
    h->n = len;
    *out = h;
    API_CATCH
}
int32_t dsa_vec_create_empty(dsa_vec_t** out) { return dsa_vec_create(nullptr, nullptr, 0, DSA_COMBINE_ADD, -1, out); }
int32_t dsa_vec_destroy(dsa_vec_t* h) { if (h) { pma_destroy(h->P); delete h; } return DSA_OK; }

static void vec_flush(dsa_vec_t* h);

int32_t dsa_vec_get_batch(dsa_vec_t* h, const int64_t* keys, int64_t n, double* out) {
    API_TRY vec_flush(h); get_batch(h->P, 0, keys, nullptr, n, out); API_CATCH
}
int32_t dsa_vec_get(dsa_vec_t* h, int64_t key, double* out) { return dsa_vec_get_batch(h, &key, 1, out); }

static void vec_apply(dsa_vec_t* h, const int64_t* keys, const double* vals, int64_t n);
static void vec_flush(dsa_vec_t* h) {
    bind_device(h->P);
    if (h->pk.empty()) return;
    std::vector<int64_t> k; std::vector<double> v;
    k.swap(h->pk); v.swap(h->pv);                      // the queue is empty even if the apply fails
    vec_apply(h, k.data(), v.data(), (int64_t)k.size());
}
static void vec_apply(dsa_vec_t* h, const int64_t* keys, const double* vals, int64_t n) {
    int32_t err = 0;
    static const bool par = [] { const char* e = dev_env("DSA_PARBATCH"); return !(e && e[0] == '0'); }();
    int64_t done;
    if (par && n >= 128) done = run_ops_parallel(h->P, OpBatch(OP_VEC_SET, keys, nullptr, vals, n), &err);      // the caller's columns go up as they are
    else {
        std::vector<Op> ops((size_t)n);
        for (int64_t i = 0; i < n; ++i) ops[(size_t)i] = make_op(OP_VEC_SET, keys[i], 0, vals[i]);
        done = run_ops(h->P, ops, &err);
    }
    const int64_t upto = err ? std::min(done + 1, n) : done;          // v.n is updated before the write (src/vector.jl:77-79)
    for (int64_t i = 0; i < upto; ++i) if (vals[i] != 0.0) h->n = std::max(h->n, keys[i]);
    if (err) fail(err, err_text(err));
}

int32_t dsa_vec_set_batch(dsa_vec_t* h, const int64_t* keys, const double* vals, int64_t n) {
    API_TRY
    vec_flush(h);
    vec_apply(h, keys, vals, n);
    API_CATCH
}
int32_t dsa_vec_set(dsa_vec_t* h, int64_t key, double val) {      // queued; see PENDING_FLUSH
    API_TRY
    if (val != 0.0) h->n = std::max(h->n, key);
    h->pk.push_back(key); h->pv.push_back(val);
    if (h->pk.size() >= PENDING_FLUSH) vec_flush(h);
    API_CATCH
}
int32_t dsa_vec_nnz(dsa_vec_t* h, int64_t* out) { API_TRY vec_flush(h); *out = h->P.h_ctl->nb_elements; API_CATCH }
int32_t dsa_vec_len(dsa_vec_t* h, int64_t* out) { *out = h->n; return DSA_OK; }

int32_t dsa_vec_nonzeros(dsa_vec_t* h, int64_t* keys, double* vals, int64_t cap, int64_t* n_out) {
    API_TRY
    vec_flush(h);
    std::vector<int64_t> ks; std::vector<double> vs;
    read_range(h->P, 1, h->P.capacity(), ks, vs);
    if ((int64_t)ks.size() > cap) fail(DSA_ECAP, "output buffers too small");
    std::copy(ks.begin(), ks.end(), keys); std::copy(vs.begin(), vs.end(), vals);
    *n_out = (int64_t)ks.size();
    API_CATCH
}
// K-pack of the whole vector into its alternate slot buffer (free between rebalances); returns the number of stored cells.
// The count is known on return, the packed cells are still being written on the vector's OWN stream: a consumer on another stream
// (the other operand of == / +) must wait for it (wait_for_pack).
// K-pack of up to VIEW_SMALL_SLOTS slots by ONE launch, the count handed back through the pinned landing area of `P` (no tile counts, no scan, no
// copy, no stream synchronisation: 50 -> 15 us); returns -1 when the range does not qualify
static int64_t pack_small(Pma& P, KeyArr k, const double* v, const uint64_t* occ, int64_t from, int64_t to, KeyArr ok, double* ov, int64_t out_cap) {
    if (!publish_enabled() || to < from || from < 1 || to - from + 1 > VIEW_SMALL_SLOTS || to - from + 1 > out_cap) return -1;
    ViewAreaLease lease(P);
    const unsigned long long seq = ++P.view_seq;
    hipError_t e = launch_view_small(k, v, occ, nullptr, nullptr, nullptr, 0, to, 0, ok, ov, out_cap, P.d_small, P.h_view, 0, seq, from, to, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("pack launch: ") + hipGetErrorString(e));
    wait_policy_block(P);
    volatile int64_t* seqp = P.h_view + 5;
    auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    while ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) {
        if (std::chrono::steady_clock::now() < next_query) continue;
        const hipError_t q = hipStreamQuery(P.stream);
        if (q == hipErrorNotReady) { next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2); continue; }
        if (q != hipSuccess) fail(DSA_EHIP, std::string("pack: ") + hipGetErrorString(q));
        if ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) fail(DSA_EHIP, "pack kernel finished without publishing its result");
    }
    return P.h_view[4];
}
static int64_t vec_pack_alt(dsa_vec_t* h) {
    Pma& P = h->P;
    int64_t cnt = 0;
    const int alt = 1 - P.cur;
    {
        const int64_t c = pack_small(P, P.K(), P.V(), P.O(), 1, P.capacity(), P.KA(alt), P.vals[alt], P.cap_alloc);
        if (c >= 0) return c;
    }
    hipError_t e = launch_compact_range(P.K(), P.V(), P.O(), 1, P.capacity(), P.KA(alt), P.vals[alt], P.cap_alloc, &P.work, &cnt, P.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("compact launch: ") + hipGetErrorString(e));
    return cnt;
}
// the packed cells of `producer` (vec_pack_alt) are complete before anything enqueued on `consumer`'s stream afterwards runs
static void wait_for_pack(dsa_vec_t* producer, dsa_vec_t* consumer) {
    if (producer->P.stream == consumer->P.stream) return;
    HIPCHK(hipStreamSynchronize(producer->P.stream));
}
// v1 == v2  src/vector.jl:85-87 (lengths, then src/pma.jl:262-266: nb_elements, then the stored tuples pairwise in slot order,
// _arrays_equal src/pma.jl:236-260); compared on the device, only the verdict crosses PCIe
int32_t dsa_vec_equal(dsa_vec_t* a, dsa_vec_t* b, int32_t* out) {
    API_TRY
    vec_flush(a);
    if (a == b) { *out = 1; return DSA_OK; }
    vec_flush(b);
    *out = 0;
    if (a->n != b->n || a->P.h_ctl->nb_elements != b->P.h_ctl->nb_elements) return DSA_OK;
    const int64_t n = a->P.h_ctl->nb_elements;
    if (n == 0) { *out = 1; return DSA_OK; }
    if (vec_pack_alt(a) != n || vec_pack_alt(b) != n) fail(DSA_EASSERT, "stored-cell count differs from nb_elements");
    Pma &A = a->P, &B = b->P;
    wait_for_pack(b, a);
    HIPCHK(hipMemsetAsync(A.d_err, 0, sizeof(int32_t), A.stream));
    hipError_t e = launch_packed_equal(A.KA(1 - A.cur), A.vals[1 - A.cur], B.KA(1 - B.cur), B.vals[1 - B.cur], n, A.d_err, A.stream);
    if (e != hipSuccess) fail(DSA_EHIP, std::string("compare launch: ") + hipGetErrorString(e));
    int32_t differ = 0;
    HIPCHK(hipMemcpyAsync(&differ, A.d_err, sizeof(int32_t), hipMemcpyDeviceToHost, A.stream));
    HIPCHK(hipStreamSynchronize(A.stream));
    *out = differ ? 0 : 1;
    API_CATCH
}
// alpha * a + beta * b as ascending (key, value) pairs — the SparseVector the reference's v1 + v2 / v1 - v2 / -v evaluate to
// through the AbstractSparseVector fallbacks over nonzeroinds / nonzeros (src/vector.jl:93-109; test/functional/math.jl:53-94).
// Both operands are packed and merged on the device; a key stored in both keeps one entry unless its value is zero.
int32_t dsa_vec_axpby(dsa_vec_t* a, double alpha, dsa_vec_t* b, double beta, int64_t* keys, double* vals, int64_t cap, int64_t* n_out) {
    API_TRY
    vec_flush(a);
    if (b != a) vec_flush(b);
    *n_out = 0;
    const int64_t na = vec_pack_alt(a);
    const int64_t nb = b != a ? vec_pack_alt(b) : na;
    const int64_t total = na + nb;
    if (total == 0) return DSA_OK;
    Pma &A = a->P, &B = b->P;
    if (b != a) wait_for_pack(b, a);
    const int64_t nwords = (total + 63) >> 6, ntiles = (nwords + 63) / 64;
    char* scratch = nullptr;
    const size_t off_mv = (size_t)total * 8, off_ok = 2 * off_mv, off_ov = 3 * off_mv, off_keep = 4 * off_mv,
                 off_cnt = off_keep + (size_t)nwords * 8, off_off = off_cnt + (size_t)(ntiles + 8) * 4, bytes = off_off + (size_t)(ntiles + 8) * 4;
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&scratch), bytes));      // (the caching allocator: a hipMalloc + hipFree pair per call cost more than the kernels)
    try {
        int64_t* mk = reinterpret_cast<int64_t*>(scratch);
        double* mv = reinterpret_cast<double*>(scratch + off_mv);
        int64_t* ok = reinterpret_cast<int64_t*>(scratch + off_ok);
        double* ov = reinterpret_cast<double*>(scratch + off_ov);
        uint64_t* keep = reinterpret_cast<uint64_t*>(scratch + off_keep);
        RebalanceWork work{reinterpret_cast<uint32_t*>(scratch + off_cnt), reinterpret_cast<uint32_t*>(scratch + off_off), ntiles + 8};
        HIPCHK(hipMemsetAsync(keep, 0, (size_t)nwords * 8, A.stream));
        hipError_t e = launch_merge_axpby(A.KA(1 - A.cur), A.vals[1 - A.cur], na, alpha, B.KA(1 - B.cur), B.vals[1 - B.cur], nb, beta,
                                          mk, mv, keep, A.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("merge launch: ") + hipGetErrorString(e));
        int64_t cnt = pack_small(A, KeyArr{mk, 1, 0}, mv, keep, 1, total, KeyArr{ok, 1, 0}, ov, total);
        if (cnt < 0) {
            e = launch_compact_range(KeyArr{mk, 1, 0}, mv, keep, 1, total, KeyArr{ok, 1, 0}, ov, total, &work, &cnt, A.stream);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("compact launch: ") + hipGetErrorString(e));
        }
        if (cnt > cap) fail(DSA_ECAP, "output buffers too small");
        if (cnt > 0) {
            HIPCHK(hipMemcpyAsync(keys, ok, (size_t)cnt * 8, hipMemcpyDeviceToHost, A.stream));
            HIPCHK(hipMemcpyAsync(vals, ov, (size_t)cnt * 8, hipMemcpyDeviceToHost, A.stream));
            HIPCHK(hipStreamSynchronize(A.stream));
        }
        else HIPCHK(hipStreamSynchronize(A.stream));      // the scratch goes back to the pool: nothing may still use it
        *n_out = cnt;
    } catch (...) { (void)hipStreamSynchronize(A.stream); pool_free(scratch); throw; }
    pool_free(scratch);
    API_CATCH
}
int32_t dsa_vec_shrink_size(dsa_vec_t* h) {     // shrink_size!  src/vector.jl:64 (+ _guess_length :7-8)
    API_TRY
    vec_flush(h);
    std::vector<int64_t> ks; std::vector<double> vs;
    read_range(h->P, 1, h->P.capacity(), ks, vs);
    int64_t n = 0;
    for (int64_t k : ks) n = std::max(n, k);
    h->n = n;
    API_CATCH
}
int32_t dsa_vec_info(dsa_vec_t* h, int64_t* info) { API_TRY vec_flush(h); pma_info(h->P, h->n, info); API_CATCH }
int32_t dsa_vec_export_layout(dsa_vec_t* h, int64_t* keys, double* vals, uint8_t* occ, int64_t cap) {
    API_TRY vec_flush(h); export_slots(h->P, keys, vals, occ, cap); API_CATCH
}
int32_t dsa_vec_rebalance_root(dsa_vec_t* h) {
    API_TRY
    vec_flush(h);
    Pma& P = h->P;
    if (P.capacity() != P.h_ctl->segment_capacity) {
        P.h_ctl->stat_rebalances += 1; P.h_ctl->stat_window_slots += P.capacity();
        root_rebalance(P, P.capacity(), P.capacity(), P.h_ctl->nb_elements, false);
    }
    API_CATCH
}
int32_t dsa_vec_dev_relayout(dsa_vec_t* h, int32_t mode) {
    API_TRY
    vec_flush(h);
    Pma& P = h->P;
    Ctl& c = *P.h_ctl;
    const int64_t cap = c.capacity, m = c.nb_elements;
    if (m < 1) fail(DSA_EARG, "relayout of an empty vector");
    if (mode == 1 || mode == 2) {
        root_rebalance(P, cap, m, m, false);                   // W = m: no gaps -> the cells land on slots 1..m
        if (mode == 2) {
            const int alt = 1 - P.cur;
            ++P.layout_epoch;
            hipError_t e = launch_pack_right(P.K(), P.V(), m, P.KA(alt), P.vals[alt], P.occ[alt], cap, P.stream);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("pack-right launch: ") + hipGetErrorString(e));
            P.occ_dirty[alt] = std::max<int64_t>(P.occ_dirty[alt], (cap + 63) / 64);
            P.cur = alt;
        }
    } else if (mode == 3 || mode == 4) {
        if (mode == 4 && (cap < 4 * c.segment_capacity || 2 * m > cap)) fail(DSA_EARG, "cannot shrink");
        const int64_t old_cap = cap;
        if (mode == 3) { c.capacity *= 2; c.nb_segments *= 2; c.height += 1; } else { c.capacity /= 2; c.nb_segments /= 2; c.height -= 1; }
        compute_bounds(P);
        root_rebalance(P, old_cap, c.capacity, m, false);
        // (stream-ordered like the write path's own _extend! — which re-uploads the block with its relaunch —: no host wait in the hook)
        hipError_t e = launch_store_ctl(P.d_ctl, c, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("control block store: ") + hipGetErrorString(e));
    } else fail(DSA_EARG, "mode must be 1..4");
    API_CATCH
}
int32_t dsa_vec_check(dsa_vec_t* h, int64_t* report) { API_TRY vec_flush(h); pma_check(h->P, report); API_CATCH }
int32_t dsa_vec_set_stream(dsa_vec_t* h, void* s) {
    API_TRY
    vec_flush(h);
    HIPCHK(hipStreamSynchronize(h->P.stream));
    if (h->P.own_stream) stream_put(h->P.stream, h->P.device);
    h->P.stream = (hipStream_t)s; h->P.own_stream = false;
    API_CATCH
}
int32_t dsa_vec_sync(dsa_vec_t* h) { API_TRY vec_flush(h); HIPCHK(hipStreamSynchronize(h->P.stream)); API_CATCH }
int32_t dsa_vec_set_wait_policy(dsa_vec_t* h, int32_t policy) {
    API_TRY
    if (policy != DSA_WAIT_SPIN && policy != DSA_WAIT_BLOCK) fail(DSA_EARG, "wait policy must be DSA_WAIT_SPIN or DSA_WAIT_BLOCK");
    h->P.wait_policy = policy;
    API_CATCH
}

// ---------------- PackedCSC ----------------
int32_t dsa_pcsc_create(const int64_t* colptr, int64_t nparts, const int64_t* row_keys, const double* vals,
                        int32_t combine_op, dsa_pcsc_t** out) {
    API_TRY
    if (nparts <= 0) fail(DSA_EARG, "PackedCSC needs at least one partition");
    const int64_t nnz = colptr[nparts] - colptr[0];
    if (nnz > 0xffffffffll) fail(DSA_EARG, "more than 2^32-1 entries in one call");
    std::vector<int64_t> part((size_t)nnz);      // partition id of every entry (index expansion only; the sort / combine
    for (int64_t p = 0; p < nparts; ++p)          // of src/pcsr.jl:36-51 runs on the device)
        for (int64_t e = colptr[p]; e < colptr[p + 1]; ++e) part[(size_t)(e - colptr[0])] = p + 1;
    auto* h = new dsa_pcsc();
    try {
        pma_init_common(h->P, true, false);
        pma_build_from_host(h->P, part.data(), row_keys + colptr[0], vals + colptr[0], nnz, combine_op, 2, nparts);
    } catch (...) { pma_destroy(h->P); delete h; throw; }
    *out = h;
    API_CATCH
}
int32_t dsa_pcsc_create_empty(dsa_pcsc_t** out) {
    API_TRY
    auto* h = new dsa_pcsc();
    try {
        pma_init_common(h->P, true, false);
        ensure_tables(h->P, 64);
        build_from_packed(h->P, {}, {});
    } catch (...) { pma_destroy(h->P); delete h; throw; }
    *out = h;
    API_CATCH
}
int32_t dsa_pcsc_destroy(dsa_pcsc_t* h) { if (h) { pma_destroy(h->P); delete h; } return DSA_OK; }
int32_t dsa_pcsc_get(dsa_pcsc_t* h, int64_t key, int64_t partition, double* out) {
    API_TRY bind_device(h->P); get_batch(h->P, 1, &key, &partition, 1, out); API_CATCH
}
int32_t dsa_pcsc_set(dsa_pcsc_t* h, double val, int64_t key, int64_t partition) {
    API_TRY
    bind_device(h->P);
    if (partition > h->P.h_ctl->table_len + (1 << 24)) fail(DSA_EARG, "partition index unreasonably far past the last partition");
    std::vector<Op> ops{make_op(OP_PCSC_SET, key, partition, val)};
    int32_t err = 0;
    run_ops(h->P, ops, &err);
    if (err) fail(err, err_text(err));
    API_CATCH
}
int32_t dsa_pcsc_deletepartition(dsa_pcsc_t* h, int64_t partition) {
    API_TRY
    bind_device(h->P);
    std::vector<Op> ops{make_op(OP_DELETE_PARTITION, 0, partition, 0.0)};
    int32_t err = 0;
    run_ops(h->P, ops, &err);
    if (err) fail(err, err_text(err));
    API_CATCH
}
int32_t dsa_pcsc_nnz(dsa_pcsc_t* h, int64_t* out) { *out = h->P.h_ctl->nb_elements - h->P.h_ctl->nb_partitions; return DSA_OK; }
int32_t dsa_pcsc_nbpartitions(dsa_pcsc_t* h, int64_t* out) { *out = h->P.h_ctl->nb_partitions; return DSA_OK; }
int32_t dsa_pcsc_info(dsa_pcsc_t* h, int64_t* info) { pma_info(h->P, h->P.h_ctl->nb_partitions, info); return DSA_OK; }
int32_t dsa_pcsc_export_layout(dsa_pcsc_t* h, int64_t* keys, double* vals, uint8_t* occ, int64_t cap,
                               int64_t* semaphores, int64_t table_cap) {
    API_TRY
    bind_device(h->P);
    export_slots(h->P, keys, vals, occ, cap);
    export_tables(h->P, semaphores, nullptr, nullptr, table_cap);
    API_CATCH
}

// ---------------- matrix ----------------
int32_t dsa_mat_create_from_coo(const int64_t* I, const int64_t* J, const double* V, int64_t nnz, int64_t m, int64_t n,
                                dsa_mat_t** out) {
    API_TRY
    if (nnz < 0) fail(DSA_EARG, "negative length");
    if (nnz > 0xffffffffll) fail(DSA_EARG, "more than 2^32-1 triples in one call");
    auto* h = new dsa_mat();
    KeyRange rows, cols;
    try { mat_build_major(h, I, J, V, nnz, &rows, &cols); mat_prefetch_spmv_meta(h); } catch (...) { pma_destroy(h->col); pma_destroy(h->row); delete h; throw; }
    if (m < 0) m = rows.known() ? std::max<int64_t>(0, rows.hi) : 0;        // _guess_length  src/vector.jl:6
    if (n < 0) n = cols.known() ? std::max<int64_t>(0, cols.hi) : 0;
    h->m = m; h->n = n;
    *out = h;
    API_CATCH
}
// ---------------- column-range shards (the reference is single-process) ----------------
// shard g of G owns the global column keys (col0, col0 + ncols]: n / G columns each, the first n % G shards one more
int32_t dsa_shard_range(int64_t n, int32_t nshards, int32_t shard, int64_t* col0, int64_t* ncols) {
    API_TRY
    if (n < 0 || nshards <= 0 || shard < 0 || shard >= nshards) fail(DSA_EARG, "shard index / count out of range");
    const int64_t base = n / nshards, rem = n % nshards;
    *col0 = shard * base + std::min<int64_t>(shard, rem);
    *ncols = base + (shard < rem ? 1 : 0);
    API_CATCH
}
// the shard's sub-matrix as an independent reference-layout matrix: the triples whose column lies in the shard's range, with
// LOCAL column keys 1..ncols (so x is the shard's slice of the global x); size m x ncols
int32_t dsa_shard_create_from_coo(const int64_t* I, const int64_t* J, const double* V, int64_t nnz, int64_t m, int64_t n,
                                  int32_t nshards, int32_t shard, dsa_mat_t** out) {
    int64_t col0 = 0, ncols = 0;
    if (n < 0) { g_err = "a sharded matrix needs its global column count"; return DSA_EARG; }
    const int32_t rc = dsa_shard_range(n, nshards, shard, &col0, &ncols);
    if (rc != DSA_OK) return rc;
    std::vector<int64_t> li, lj; std::vector<double> lv;
    try {
        for (int64_t k = 0; k < nnz; ++k)
            if (J[k] > col0 && J[k] <= col0 + ncols) { li.push_back(I[k]); lj.push_back(J[k] - col0); lv.push_back(V[k]); }
    } catch (const std::bad_alloc&) { g_err = "host allocation failed"; return DSA_EHIP; }
    return dsa_mat_create_from_coo(li.data(), lj.data(), lv.data(), (int64_t)li.size(), m, ncols, out);
}
// partial y (length m) of one shard: y_g = A[:, range_g] * x[range_g], x and y resident in HBM, asynchronous on the handle's
// stream; the all-reduce over the shards belongs to the host layer (RCCL through torch.distributed, one process per GPU)
int32_t dsa_shard_spmv_dev(dsa_mat_t* h, const double* d_x_local, int64_t nx, double* d_y_partial, int64_t ny) {
    return dsa_mat_spmv_dense_dev(h, 0, 0, d_x_local, nx, d_y_partial, ny);
}
// y = A x of the whole sharded matrix: the local product, then the RCCL all-reduce of the partial y over the ranks (comm.hip),
// both on the shard's stream — the one collective of the path, entirely behind the ABI (dsa_shard_spmv)
int32_t dsa_shard_spmv_allreduce_dev(dsa_mat_t* h, dsa_comm_t* comm, const double* d_x_local, int64_t nx, double* d_y, int64_t ny) {
    const int32_t rc = dsa_mat_spmv_dense_dev(h, 0, 0, d_x_local, nx, d_y, ny);
    if (rc != DSA_OK) return rc;
    return dsa_shard_allreduce_dev(comm, d_y, ny, h->row.stream);
}
int32_t dsa_mat_create_empty(int32_t fill_mode, dsa_mat_t** out) {
    API_TRY
    auto* h = new dsa_mat();
    if (fill_mode) {
        HIPCHK(hipSetDevice(g_device));     // fail loudly without a device even though nothing is allocated yet
        h->fillmode = true;
    } else {
        try { mat_build_major(h, nullptr, nullptr, nullptr, 0); } catch (...) { pma_destroy(h->col); pma_destroy(h->row); delete h; throw; }
    }
    *out = h;
    API_CATCH
}
int32_t dsa_mat_destroy(dsa_mat_t* h) {
    if (h) {
        if (h->has_major) { pma_destroy(h->col); pma_destroy(h->row); }
        fill_release(h->buf);
        if (h->d_x) hipFree(h->d_x);
        if (h->d_y) hipFree(h->d_y);
        {
            dsa_mat::Spx& x = h->spx;
            if (x.res_stream) (void)hipStreamSynchronize(x.res_stream);
            pool_free(x.acc); pool_free(x.bm); pool_free(x.tile_cnt); pool_free(x.tile_off); pool_free(x.ticket);
            pool_free(x.oi); pool_free(x.ov); pool_free(x.dx); pool_free(x.d_count);
            pinned_free(x.pin);
            for (hipEvent_t e : x.ev) if (e) (void)hipEventDestroy(e);
            if (x.stage) (void)hipHostFree(x.stage);
        }
        delete h;
    }
    return DSA_OK;
}

// ---- device-resident fill buffer --------------------------------------------------------------------------------------
static void fill_init(FillBuffer& b) {
    if (b.stream) return;
    HIPCHK(hipSetDevice(g_device));
    b.device = g_device;
    HIPCHK(stream_get(&b.stream));
    {
        std::lock_guard<std::mutex> lk(g_fill_cache.mu);
        if (g_fill_cache.hI[0] != nullptr && g_fill_cache.device == b.device)
            for (int k = 0; k < 2; ++k) { b.hI[k] = g_fill_cache.hI[k]; b.hJ[k] = g_fill_cache.hJ[k]; b.hV[k] = g_fill_cache.hV[k]; g_fill_cache.hI[k] = nullptr; g_fill_cache.hJ[k] = nullptr; g_fill_cache.hV[k] = nullptr; }
    }
    {
        void* p = nullptr;
        HIPCHK(pool_alloc(&p, 64)); b.d_acc = static_cast<long long*>(p);
        HIPCHK(pinned_alloc(&p, 64)); b.h_acc = static_cast<long long*>(p);
        b.h_acc[0] = INT64_MAX; b.h_acc[1] = INT64_MIN; b.h_acc[2] = INT64_MAX; b.h_acc[3] = INT64_MIN; b.h_acc[4] = 0;
        HIPCHK(hipMemcpyAsync(b.d_acc, b.h_acc, 5 * sizeof(long long), hipMemcpyHostToDevice, b.stream));
        HIPCHK(hipStreamSynchronize(b.stream));             // (h_acc is reused as the read-back target)
    }
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipEventCreateWithFlags(&b.uploaded[k], hipEventDisableTiming));
        if (b.hI[k]) continue;
        HIPCHK(hipHostMalloc(&b.hI[k], (size_t)FillBuffer::CHUNK * sizeof(int64_t), hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&b.hJ[k], (size_t)FillBuffer::CHUNK * sizeof(int64_t), hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&b.hV[k], (size_t)FillBuffer::CHUNK * sizeof(double), hipHostMallocDefault));
    }
}
// ships what is staged in the current pinned chunk and not yet sent (triples [b.sent, b.fill)) to HBM, asynchronously; a chunk that
// is full (or `finish`: the end of a batch) is closed — its event recorded — and the other chunk becomes current.  Pieces go up
// while the caller's memcpy fills the rest of the chunk, so that closefillmode! waits for one piece (6 MB), not for a chunk (24 MB).
static void fill_upload_chunk(FillBuffer& b, bool finish = true) {
    if (b.fill == 0) return;
    HIPCHK(hipSetDevice(b.device));
    const int64_t npiece = b.fill - b.sent;
    if (npiece > 0) {
        if (b.dlen + npiece > b.dcap) {          // grow geometrically: new arrays, device-to-device copy of what is resident
            const int64_t ncap = std::max<int64_t>(2 * b.dcap, std::max<int64_t>(b.dlen + npiece, 4 * FillBuffer::CHUNK));
            int64_t *nI = nullptr, *nJ = nullptr; double* nV = nullptr;
            HIPCHK(pool_alloc(reinterpret_cast<void**>(&nI), (size_t)ncap * sizeof(int64_t)));
            HIPCHK(pool_alloc(reinterpret_cast<void**>(&nJ), (size_t)ncap * sizeof(int64_t)));
            HIPCHK(pool_alloc(reinterpret_cast<void**>(&nV), (size_t)ncap * sizeof(double)));
            if (b.dlen > 0) {
                HIPCHK(hipMemcpyAsync(nI, b.dI, (size_t)b.dlen * sizeof(int64_t), hipMemcpyDeviceToDevice, b.stream));
                HIPCHK(hipMemcpyAsync(nJ, b.dJ, (size_t)b.dlen * sizeof(int64_t), hipMemcpyDeviceToDevice, b.stream));
                HIPCHK(hipMemcpyAsync(nV, b.dV, (size_t)b.dlen * sizeof(double), hipMemcpyDeviceToDevice, b.stream));
            }
            HIPCHK(hipStreamSynchronize(b.stream));       // earlier uploads into the old arrays have landed
            pool_free(b.dI); pool_free(b.dJ); pool_free(b.dV);
            b.dI = nI; b.dJ = nJ; b.dV = nV; b.dcap = ncap;
        }
        const int c = b.cur;
        HIPCHK(hipMemcpyAsync(b.dI + b.dlen, b.hI[c] + b.sent, (size_t)npiece * sizeof(int64_t), hipMemcpyHostToDevice, b.stream));
        HIPCHK(hipMemcpyAsync(b.dJ + b.dlen, b.hJ[c] + b.sent, (size_t)npiece * sizeof(int64_t), hipMemcpyHostToDevice, b.stream));
        HIPCHK(hipMemcpyAsync(b.dV + b.dlen, b.hV[c] + b.sent, (size_t)npiece * sizeof(double), hipMemcpyHostToDevice, b.stream));
        {   // the value ranges of the piece, folded into the running ones behind its upload
            hipError_t e = launch_key_scan_acc(b.dI + b.dlen, b.dJ + b.dlen, npiece, b.d_acc, b.stream);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("key range launch: ") + hipGetErrorString(e));
        }
        b.dlen += npiece;
        b.sent = b.fill;
    }
    if (!finish && b.fill < FillBuffer::CHUNK) return;
    const int c = b.cur;
    HIPCHK(hipEventRecord(b.uploaded[c], b.stream));
    b.in_flight[c] = true;
    b.fill = 0; b.sent = 0;
    b.cur = 1 - c;
    if (b.in_flight[b.cur]) { HIPCHK(hipEventSynchronize(b.uploaded[b.cur])); b.in_flight[b.cur] = false; }   // its pinned memory is free again
}
// addelem!  src/buffer.jl:20-31 for n entries (n = 1: one setindex! in fill mode)
static void fill_append(FillBuffer& b, const int64_t* I, const int64_t* J, const double* V, int64_t n) {
    fill_init(b);
    int64_t k = 0;
    while (k < n) {
        const int64_t room = std::min(FillBuffer::CHUNK - b.fill, FillBuffer::PIECE - (b.fill - b.sent));      // up to the end of the chunk / of the piece
        const int64_t take = std::min(room, n - k);
        std::memcpy(b.hI[b.cur] + b.fill, I + k, (size_t)take * sizeof(int64_t));
        std::memcpy(b.hJ[b.cur] + b.fill, J + k, (size_t)take * sizeof(int64_t));
        std::memcpy(b.hV[b.cur] + b.fill, V + k, (size_t)take * sizeof(double));
        b.fill += take; k += take;
        if (b.fill == FillBuffer::CHUNK || b.fill - b.sent >= FillBuffer::PIECE) fill_upload_chunk(b, false);
    }
    // a batch of appends has ended: what is staged goes up now (asynchronously), so that closefillmode! finds almost nothing left
    // in pinned memory — single-element appends (setindex! in fill mode) only ship whole quarter chunks
    if (b.fill - b.sent >= (n > 1 ? FillBuffer::EAGER : FillBuffer::CHUNK / 4)) fill_upload_chunk(b, false);
    b.length += n;
}

static void mat_flush(dsa_mat_t* h) {
    if (h->has_major) bind_device(h->col);
    if (h->pi.empty()) return;
    std::vector<int64_t> i, j; std::vector<double> v;
    i.swap(h->pi); j.swap(h->pj); v.swap(h->pv);       // the queue is empty even if the apply fails
    mat_apply_sets(h, i.data(), j.data(), v.data(), (int64_t)i.size());
    mat_prefetch_spmv_meta(h);
}

int32_t dsa_mat_set(dsa_mat_t* h, double val, int64_t row, int64_t col) {
    API_TRY
    check_key(row); check_key(col);
    if (val != 0.0) { h->m = std::max(h->m, row); h->n = std::max(h->n, col); }      // src/matrix.jl:44-47
    if (h->fillmode) {
        fill_row_test_and_set(h->buf, row);
        fill_append(h->buf, &row, &col, &val, 1);
    } else {
        h->pi.push_back(row); h->pj.push_back(col); h->pv.push_back(val);
        // with tombstones a write can hit the reference's assert / bounds paths: apply it now so the error surfaces here
        const bool tombstones = h->col.h_ctl->nb_partitions != h->col.h_ctl->table_len ||
                                h->row.h_ctl->nb_partitions != h->row.h_ctl->table_len;
        if (tombstones || h->pi.size() >= PENDING_FLUSH) mat_flush(h);
    }
    API_CATCH
}

int32_t dsa_mat_set_batch(dsa_mat_t* h, const int64_t* I, const int64_t* J, const double* V, int64_t n) {
    API_TRY
    for (int64_t k = 0; k < n; ++k) { check_key(I[k]); check_key(J[k]); }
    mat_flush(h);
    if (h->fillmode) {
        for (int64_t k = 0; k < n; ++k) {
            if (V[k] != 0.0) { h->m = std::max(h->m, I[k]); h->n = std::max(h->n, J[k]); }
            fill_row_test_and_set(h->buf, I[k]);
        }
        fill_append(h->buf, I, J, V, n);
    } else {
        mat_apply_sets(h, I, J, V, n);
        mat_prefetch_spmv_meta(h);
    }
    API_CATCH
}

int32_t dsa_mat_get_batch(dsa_mat_t* h, const int64_t* I, const int64_t* J, int64_t n, double* out) {
    API_TRY
    mat_flush(h);
    if (h->fillmode) fail(DSA_EMODE, "getindex(row, col) is not available in fill mode.");
    get_batch(h->col, 2, I, J, n, out);
    API_CATCH
}
int32_t dsa_mat_get(dsa_mat_t* h, int64_t row, int64_t col, double* out) { return dsa_mat_get_batch(h, &row, &col, 1, out); }

int32_t dsa_mat_addrow(dsa_mat_t* h, int64_t row, const int64_t* colids, const double* vals, int64_t n) {
    API_TRY
    mat_flush(h);
    check_key(row);
    for (int64_t k = 0; k < n; ++k) check_key(colids[k]);
    if (h->fillmode) {     // addrow!(buffer, ...)  src/buffer.jl:10-18
        FillBuffer& b = h->buf;
        if (fill_row_test_and_set(b, row)) fail(DSA_EMODE, "Row already written in dynamic sparse matrix buffer.");
        std::vector<int64_t> rows((size_t)n, row);       // (the reference stores the column ids of the row sorted: only its buffer views see that)
        fill_append(b, rows.data(), colids, vals, n);
    } else {               // src/matrix.jl:119-121
        std::vector<int64_t> rows((size_t)n, row);
        mat_apply_sets(h, rows.data(), colids, vals, n);
        mat_prefetch_spmv_meta(h);
    }
    API_CATCH
}

int32_t dsa_mat_closefillmode(dsa_mat_t* h) {     // closefillmode!  src/matrix.jl:126-134
    API_TRY
    mat_flush(h);
    if (!h->fillmode) fail(DSA_EMODE, "Cannot close fill mode because matrix is not in fill mode.");
    // get_rowids_colids_vals (src/buffer.jl:33-50) is a no-op here: the triples already sit in HBM; only the last partial
    // chunk is still in pinned memory
    FillBuffer& b = h->buf;
    int64_t nnz = 0;
    bool wr = false, wc = false;
    static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    // value ranges of everything appended (K-build's composite, the storage width of the keys): every uploaded piece was folded into
    // five running words on the device (k_minmax_acc) — they come back with the wait for the last piece; the appends themselves never
    // look at a key twice
    KeyRange rows, cols;
    if (b.stream) {
        fill_upload_chunk(b);
        HIPCHK(hipMemcpyAsync(b.h_acc, b.d_acc, 5 * sizeof(long long), hipMemcpyDeviceToHost, b.stream));
        HIPCHK(hipStreamSynchronize(b.stream));
        nnz = b.dlen;
        if (nnz > 0) {
            rows.lo = b.h_acc[0]; rows.hi = b.h_acc[1]; cols.lo = b.h_acc[2]; cols.hi = b.h_acc[3];
            wr = g_force_wide || !(key_fits32(rows.lo) && key_fits32(rows.hi));
            wc = g_force_wide || !(key_fits32(cols.lo) && key_fits32(cols.hi));
        }
    }
    // a failed build (out of memory, a HIP error) leaves the matrix what it was: in fill mode, with all of its triples — the
    // builder only reads them — so the caller may free memory and close again, or keep appending
    const auto tc1 = std::chrono::steady_clock::now();
    mat_build_major_dev(h, b.dI, b.dJ, b.dV, nnz, wr, wc, rows, cols);
    const auto tc2 = std::chrono::steady_clock::now();
    h->fillmode = false;
    fill_release(h->buf);
    mat_prefetch_spmv_meta(h);
    if (dbg_time) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) { return std::chrono::duration<double, std::milli>(b2 - a).count(); };
        fprintf(stderr, "[closefillmode] last chunk %.2f ms  build %.2f ms  release + prefetch %.2f ms\n", ms(tc0, tc1), ms(tc1, tc2), ms(tc2, std::chrono::steady_clock::now()));
    }
    API_CATCH
}

int32_t dsa_mat_deletecolumn(dsa_mat_t* h, int64_t col) {      // src/matrix.jl:95-102
    API_TRY
    mat_flush(h);
    if (h->fillmode) fail(DSA_EMODE, "Cannot delete a column in fill mode");
    static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
    const auto td0 = std::chrono::steady_clock::now();
    std::vector<int64_t> rows; std::vector<double> vals;
    col_view_of(h->col, col, rows, vals);
    const auto td1 = std::chrono::steady_clock::now();
    std::vector<Op> ops;
    for (int64_t r : rows) ops.push_back(make_op(OP_MPCSC_SET, col, r, 0.0));     // rowmajor[col, row] = 0
    std::vector<Op> del{make_op(OP_MPCSC_DELETECOLUMN, 0, col, 0.0)};
    struct Report { bool on; std::chrono::steady_clock::time_point a, b; size_t n; ~Report() { if (on) fprintf(stderr, "[deletecolumn] view %.1f us, %zu twin deletes + deletepartition %.1f us\n",
        std::chrono::duration<double, std::micro>(b - a).count(), n, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - b).count()); } } report{dbg_time, td0, td1, rows.size()};
    // the element deletes of the twin cannot fail and touch the other structure: the deletepartition! of the own orientation is launched
    // on its sequencer, the twin's deletes — different partitions, mostly disjoint footprints — go through the local rounds meanwhile
    // (one wave per op instead of one op after the other: 105 -> 60 us for a column of 16)
    static const bool twin_rounds = [] { const char* e = dev_env("DSA_TWIN_ROUNDS"); return !(e && e[0] == '0'); }();
    if (h->col.stream != h->row.stream && twin_rounds && !ops.empty()) {
        SeqRun rc;
        seq_start(rc, h->col, del);
        int32_t er = 0;
        try { run_ops_parallel(h->row, ops, &er); } catch (...) { while (seq_step(rc)) {} throw; }
        while (seq_step(rc)) {}
        if (er) fail(er, err_text(er));
        if (rc.err) fail(rc.err, err_text(rc.err));
    } else if (h->col.stream != h->row.stream) {
        SeqRun rr, rc;
        run_ops_pair(h->row, ops, h->col, del, rr, rc);
        if (rr.err) fail(rr.err, err_text(rr.err));
        if (rc.err) fail(rc.err, err_text(rc.err));
    } else {
        int32_t err = 0;
        run_ops(h->row, ops, &err);
        if (err) fail(err, err_text(err));
        run_ops(h->col, del, &err);
        if (err) fail(err, err_text(err));
    }
    API_CATCH
}
int32_t dsa_mat_deleterow(dsa_mat_t* h, int64_t row) {         // src/matrix.jl:104-111
    API_TRY
    mat_flush(h);
    if (h->fillmode) fail(DSA_EMODE, "Cannot delete a row in fill mode");
    std::vector<int64_t> cols; std::vector<double> vals;
    col_view_of(h->row, row, cols, vals);
    std::vector<Op> ops;
    for (int64_t c : cols) ops.push_back(make_op(OP_MPCSC_SET, row, c, 0.0));     // colmajor[row, col] = 0
    std::vector<Op> del{make_op(OP_MPCSC_DELETECOLUMN, 0, row, 0.0)};
    static const bool twin_rounds = [] { const char* e = dev_env("DSA_TWIN_ROUNDS"); return !(e && e[0] == '0'); }();
    if (h->col.stream != h->row.stream && twin_rounds && !ops.empty()) {       // (as in deletecolumn!)
        SeqRun rr;
        seq_start(rr, h->row, del);
        int32_t ec = 0;
        try { run_ops_parallel(h->col, ops, &ec); } catch (...) { while (seq_step(rr)) {} throw; }
        while (seq_step(rr)) {}
        if (ec) fail(ec, err_text(ec));
        if (rr.err) fail(rr.err, err_text(rr.err));
    } else if (h->col.stream != h->row.stream) {
        SeqRun rc, rr;
        run_ops_pair(h->col, ops, h->row, del, rc, rr);
        if (rc.err) fail(rc.err, err_text(rc.err));
        if (rr.err) fail(rr.err, err_text(rr.err));
    } else {
        int32_t err = 0;
        run_ops(h->col, ops, &err);
        if (err) fail(err, err_text(err));
        run_ops(h->row, del, &err);
        if (err) fail(err, err_text(err));
    }
    API_CATCH
}

static void _check_status(int32_t rc) { if (rc != DSA_OK) fail(rc, g_err); }

static int32_t view_impl(dsa_mat_t* h, int32_t o, int64_t key, int64_t* ks, double* vs, int64_t cap, int64_t* n_out) {
    API_TRY
    mat_flush(h);
    if (h->fillmode) fail(DSA_EMODE, "View not available in fill mode.");
    std::vector<int64_t> k; std::vector<double> v;
    col_view_of(orient(h, o), key, k, v);
    if ((int64_t)k.size() > cap) fail(DSA_ECAP, "output buffers too small");
    std::copy(k.begin(), k.end(), ks); std::copy(v.begin(), v.end(), vs);
    *n_out = (int64_t)k.size();
    API_CATCH
}
int32_t dsa_mat_col_view(dsa_mat_t* h, int64_t col, int64_t* rows, double* vals, int64_t cap, int64_t* n_out) {
    return view_impl(h, DSA_COLMAJOR, col, rows, vals, cap, n_out);
}
int32_t dsa_mat_row_view(dsa_mat_t* h, int64_t row, int64_t* cols, double* vals, int64_t cap, int64_t* n_out) {
    return view_impl(h, DSA_ROWMAJOR, row, cols, vals, cap, n_out);
}
// m[:, col] (src/pcsr.jl:285-291 -> :247-259) and m[row, :] (src/pcsr.jl:269-283; served from the rowmajor twin, whose
// partition `row` holds exactly the (col, value) pairs the reference collects by scanning the colmajor array)
// PackedMemoryArray(elements) (src/pma.jl:69-84) of the n cells packed at the front of S's alternate buffer (view_dev: ascending,
// distinct keys — nothing to sort or fold): geometry on the host, then ONE k_move2<PACKED> from S's buffer straight into the new
// vector's slot array.  The cells never leave HBM.  Everything is enqueued on S's stream (the pack that produced the cells runs
// there); the new vector gets its own stream back before it is handed out, after the one wait of upload_ctl.
static dsa_vec* vec_from_packed_dev(Pma& S, int64_t n, int64_t len) {
    auto* v = new dsa_vec();
    hipStream_t own = nullptr;
    try {
        pma_init_common(v->P, false, false);
        Pma& P = v->P;
        own = P.stream; P.stream = S.stream;
        P.wide = S.wide;
        const int64_t capacity = capacity_for(n);
        set_geometry_for_new(P, capacity, n);
        ensure_capacity_alloc(P, 2 * capacity, false);
        ++P.layout_epoch; ++P.stat_grid_rebalances;
        P.h_ctl->stat_rebalances = 0; P.h_ctl->stat_window_slots = 0;
        if (capacity != P.h_ctl->segment_capacity) { P.h_ctl->stat_rebalances = 1; P.h_ctl->stat_window_slots = capacity; }
        // TWO launches make the vector: (1) both bitmaps, the status table of the grid rebalance and the control block (passed by value:
        // no copy command, the pinned mirror is not a DMA source) — five stream commands and a wait until round 6; (2) the spread
        hipError_t e = launch_init_fresh(P.occ[0], P.occ[1], P.occ_words, P.work.status, P.work.status_cap, P.d_ctl, *P.h_ctl, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("init launch: ") + hipGetErrorString(e));
        const int salt = 1 - S.cur;
        // (root_rebalance of P with a foreign source: cells of S.alt[1..n] -> P.cur[1..capacity])
        e = launch_rebalance(S.KA(salt), S.vals[salt], S.occ[salt], 1, n, true, P.K(), P.V(), P.O(), 1, capacity, n,
                             nullptr, &P.work, P.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("rebalance launch: ") + hipGetErrorString(e));
        P.occ_dirty[P.cur] = (capacity + 63) / 64;
        // the vector's own stream waits (on the device) for what was enqueued on S's; whatever S does next with its alternate buffer is
        // stream-ordered behind the spread.  No host wait.
        if (S.ev_handoff == nullptr) HIPCHK(hipEventCreateWithFlags(&S.ev_handoff, hipEventDisableTiming));
        HIPCHK(hipEventRecord(S.ev_handoff, S.stream));
        HIPCHK(hipStreamWaitEvent(own, S.ev_handoff, 0));
        P.stream = own;
    } catch (...) {
        if (own) { (void)hipStreamSynchronize(v->P.stream); v->P.stream = own; }
        pma_destroy(v->P); delete v; throw;
    }
    v->n = len;
    return v;
}
static int32_t slice_impl(dsa_mat_t* h, int32_t o, int64_t key, dsa_vec_t** out) {
    API_TRY
    mat_flush(h);
    if (h->fillmode) fail(DSA_EMODE, "slices are not available in fill mode");
    Pma& S = orient(h, o);
    static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
    const auto ts0 = std::chrono::steady_clock::now();
    const DevView dv = view_dev(S, key);
    const auto ts1 = std::chrono::steady_clock::now();
    if (dv.cnt <= 0) { _check_status(dsa_vec_create(nullptr, nullptr, 0, DSA_COMBINE_ADD, 0, out)); return DSA_OK; }      // PackedMemoryArray(L, T)  src/pcsr.jl:288
    *out = vec_from_packed_dev(S, dv.cnt, std::max<int64_t>(dv.last_key, 0));          // _guess_length(pma)  src/vector.jl:7-8
    if (dbg_time) fprintf(stderr, "[slice] %lld cells: view %.1f us, new vector %.1f us\n", (long long)dv.cnt,
                          std::chrono::duration<double, std::micro>(ts1 - ts0).count(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ts1).count());
    API_CATCH
}
// @view m[:, col] / @view m[row, :] with the cells delivered into HBM: d_keys / d_vals are DEVICE arrays of `cap` entries; the copy is
// enqueued on the orientation's stream (dsa_mat_set_stream / dsa_mat_sync), the count is known on return
static int32_t view_dev_impl(dsa_mat_t* h, int32_t o, int64_t key, int64_t* d_keys, double* d_vals, int64_t cap, int64_t* n_out) {
    API_TRY
    mat_flush(h);
    if (h->fillmode) fail(DSA_EMODE, "View not available in fill mode.");
    Pma& S = orient(h, o);
    const DevView dv = view_dev(S, key);
    *n_out = 0;
    if (dv.cnt <= 0) return DSA_OK;
    if (dv.cnt > cap) fail(DSA_ECAP, "output buffers too small");
    const int salt = 1 - S.cur;
    if (S.wide) HIPCHK(hipMemcpyAsync(d_keys, S.keys[salt], (size_t)dv.cnt * sizeof(int64_t), hipMemcpyDeviceToDevice, S.stream));
    else {
        hipError_t e = launch_widen_keys(S.keys[salt], d_keys, dv.cnt, S.stream);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("view copy-out: ") + hipGetErrorString(e));
    }
    HIPCHK(hipMemcpyAsync(d_vals, S.vals[salt], (size_t)dv.cnt * sizeof(double), hipMemcpyDeviceToDevice, S.stream));
    *n_out = dv.cnt;
    API_CATCH
}
int32_t dsa_mat_col_view_dev(dsa_mat_t* h, int64_t col, int64_t* d_rows, double* d_vals, int64_t cap, int64_t* n_out) {
    return view_dev_impl(h, DSA_COLMAJOR, col, d_rows, d_vals, cap, n_out);
}
int32_t dsa_mat_row_view_dev(dsa_mat_t* h, int64_t row, int64_t* d_cols, double* d_vals, int64_t cap, int64_t* n_out) {
    return view_dev_impl(h, DSA_ROWMAJOR, row, d_cols, d_vals, cap, n_out);
}
int32_t dsa_mat_col_slice(dsa_mat_t* h, int64_t col, dsa_vec_t** out) { return slice_impl(h, DSA_COLMAJOR, col, out); }
int32_t dsa_mat_row_slice(dsa_mat_t* h, int64_t row, dsa_vec_t** out) { return slice_impl(h, DSA_ROWMAJOR, row, out); }

int32_t dsa_mat_nnz(dsa_mat_t* h, int64_t* out) {      // nnz(m) = nnz(m.rowmajor)  src/matrix.jl:91
    API_TRY mat_flush(h); Pma& P = orient(h, DSA_ROWMAJOR); *out = P.h_ctl->nb_elements - P.h_ctl->nb_partitions; API_CATCH
}
int32_t dsa_mat_size(dsa_mat_t* h, int64_t* m, int64_t* n) { *m = h->m; *n = h->n; return DSA_OK; }
int32_t dsa_mat_nbpartitions(dsa_mat_t* h, int32_t o, int64_t* out) { API_TRY mat_flush(h); *out = orient(h, o).h_ctl->nb_partitions; API_CATCH }
int32_t dsa_mat_info(dsa_mat_t* h, int32_t o, int64_t* info) { API_TRY mat_flush(h); Pma& P = orient(h, o); pma_info(P, P.h_ctl->nb_partitions, info); API_CATCH }
int32_t dsa_mat_export_layout(dsa_mat_t* h, int32_t o, int64_t* keys, double* vals, uint8_t* occ, int64_t cap,
                              int64_t* semaphores, int64_t* col_keys, uint8_t* col_live, int64_t table_cap) {
    API_TRY
    mat_flush(h);
    Pma& P = orient(h, o);
    export_slots(P, keys, vals, occ, cap);
    export_tables(P, semaphores, col_keys, col_live, table_cap);
    API_CATCH
}
int32_t dsa_mat_rebalance_root(dsa_mat_t* h, int32_t o) {
    API_TRY
    mat_flush(h);
    Pma& P = orient(h, o);
    if (P.capacity() != P.h_ctl->segment_capacity) {
        P.h_ctl->stat_rebalances += 1; P.h_ctl->stat_window_slots += P.capacity();
        root_rebalance(P, P.capacity(), P.capacity(), P.h_ctl->nb_elements, false);
    }
    API_CATCH
}

int32_t dsa_mat_spmv_dense_dev(dsa_mat_t* h, int32_t transpose, int32_t algo, const double* d_x, int64_t nx, double* d_y, int64_t ny) {
    API_TRY
    mat_flush(h);
    Pma& P = transpose ? h->col : h->row;
    spmv_dev(h, transpose, algo, d_x, nx, d_y, ny, P.stream);
    API_CATCH
}

int32_t dsa_mat_spmv_dense(dsa_mat_t* h, int32_t transpose, const double* x, int64_t nx, double* y, int64_t ny) {
    API_TRY
    mat_flush(h);
    if (!h->has_major) fail(DSA_EMODE, "matrix is in fill mode");
    if (nx < 0 || ny < 0) fail(DSA_EARG, "negative length");
    ensure_xy(h, nx, ny);
    Pma& P = transpose ? h->col : h->row;
    if (nx > 0) HIPCHK(hipMemcpyAsync(h->d_x, x, (size_t)nx * sizeof(double), hipMemcpyHostToDevice, P.stream));
    if (ny > 0) {
        spmv_dev(h, transpose, 0, h->d_x, nx, h->d_y, ny, P.stream);
        HIPCHK(hipMemcpyAsync(y, h->d_y, (size_t)ny * sizeof(double), hipMemcpyDeviceToHost, P.stream));
    }
    HIPCHK(hipStreamSynchronize(P.stream));
    API_CATCH
}

// ---- sparse x (the product Coluna calls): the touched rows, ascending, stored zeros kept (_mul_output, src/operations.jl:11-12) ----
// Two device strategies (sparsex.hip), one result form:
//   few stored entries : k_spx_accum over the reference's own orientation (colmajor for mat * v) — work ~ matched cells
//   many               : densify x, gather kernel over the twin for the values + once more on the 0/1 pattern of x
// then count + emit of the touched rows from a bitmap.  dsa_mat_spmv_sparse_begin computes and leaves the packed result with the
// handle (HBM; short results also in a pinned landing area the emit kernel writes to directly), dsa_mat_spmv_sparse_fetch copies it
// out: the caller allocates exactly what the product needs (until round 6 the wrappers guessed a capacity and REPEATED the whole
// product on DSA_ECAP: four products for one at 394 k stored entries).
}  // extern "C"
namespace {
int P_device_of(dsa_mat* h) { return h->col.device; }
constexpr int64_t SPX_PIN_CELLS = 4096;          // result pairs the emit kernel hands over through pinned memory (64 KB)
constexpr int64_t SPX_DIRECT_X = 4096;           // stored entries read by k_spx_accum straight from pinned host memory (no copy command)

// host copy between a pinned staging area and the caller's pageable array: above 1 MB on up to four threads (first-touch page faults
// of a freshly allocated result array are most of the cost, and they parallelise)
void par_memcpy(void* dst, const void* src, size_t bytes) {
    if (bytes < ((size_t)1 << 20)) { std::memcpy(dst, src, bytes); return; }
    const int nt = bytes >= ((size_t)4 << 20) ? 4 : 2;
    const size_t part = ((bytes / nt) + 4095) & ~(size_t)4095;
    std::thread th[3];
    int started = 0;
    for (int t = 1; t < nt; ++t) {
        const size_t off = (size_t)t * part;
        if (off >= bytes) break;
        const size_t b = std::min(part, bytes - off);
        th[started++] = std::thread([=] { std::memcpy((char*)dst + off, (const char*)src + off, b); });
    }
    std::memcpy(dst, src, std::min(part, bytes));
    for (int t = 0; t < started; ++t) th[t].join();
}
void spx_ensure(dsa_mat* h, int64_t ny, int64_t nx_upload, hipStream_t s) {
    dsa_mat::Spx& x = h->spx;
    if (ny > x.rows_cap) {
        if (x.res_stream) HIPCHK(hipStreamSynchronize(x.res_stream));
        pool_free(x.acc); pool_free(x.bm); pool_free(x.oi); pool_free(x.ov); pool_free(x.tile_cnt); pool_free(x.tile_off);
        x.acc = nullptr; x.bm = nullptr; x.oi = nullptr; x.ov = nullptr; x.tile_cnt = nullptr; x.tile_off = nullptr; x.rows_cap = 0; x.out_cap = 0; x.tiles_cap = 0;
        const int64_t rows = std::max<int64_t>(ny + ny / 4, 4096);
        const int64_t nwords = (rows + 63) >> 6, ntiles = (nwords + 63) / 64;
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.acc), (size_t)rows * sizeof(double)));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.bm), (size_t)(ntiles * 64) * sizeof(uint64_t)));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.oi), (size_t)rows * sizeof(int64_t)));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.ov), (size_t)rows * sizeof(double)));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.tile_cnt), (size_t)(ntiles + 1) * sizeof(uint32_t)));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.tile_off), (size_t)(ntiles + 2) * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(x.acc, 0, (size_t)rows * sizeof(double), s));               // the zero invariant starts here
        HIPCHK(hipMemsetAsync(x.bm, 0, (size_t)(ntiles * 64) * sizeof(uint64_t), s));
        x.rows_cap = rows; x.out_cap = rows; x.tiles_cap = ntiles;
    }
    if (!x.ticket) {
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.ticket), 4 * sizeof(unsigned int)));
        HIPCHK(hipMemsetAsync(x.ticket, 0, 4 * sizeof(unsigned int), s));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.d_count), sizeof(int64_t)));
        HIPCHK(pinned_alloc(reinterpret_cast<void**>(&x.pin), (size_t)(8 + 2 * SPX_PIN_CELLS) * sizeof(long long)));
        std::memset(x.pin, 0, 8 * sizeof(long long));
    }
    if (nx_upload > x.x_cap) {
        if (x.res_stream) HIPCHK(hipStreamSynchronize(x.res_stream));
        pool_free(x.dx); x.dx = nullptr; x.x_cap = 0;
        const int64_t c = nx_upload + nx_upload / 2;
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&x.dx), (size_t)c * 16));
        x.x_cap = c;
    }
}
void spx_stage(dsa_mat* h, size_t bytes, hipStream_t s) {
    dsa_mat::Spx& x = h->spx;
    if (bytes <= x.stage_bytes) return;
    if (x.stage) { HIPCHK(hipStreamSynchronize(s)); if (x.res_stream && x.res_stream != s) HIPCHK(hipStreamSynchronize(x.res_stream)); HIPCHK(hipHostFree(x.stage)); }
    x.stage = nullptr; x.stage_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1u << 16);
    HIPCHK(hipHostMalloc(&x.stage, want, hipHostMallocDefault));
    x.stage_bytes = want;
}
// which strategy: the x-driven kernel costs ~ the stored entries (a handful of dependent round trips per entry, one wave each), the
// gather kernel ~ the slot array (twice: values, pattern).  DSA_SPX_XDRIVEN=0/1 forces one (A/B, coverage).
bool spx_xdriven(int64_t nx, int64_t ncols) {
    static const int force = [] { const char* e = dev_env("DSA_SPX_XDRIVEN"); return e ? atoi(e) : -1; }();
    if (force == 0 || force == 1) return force == 1;
    return nx * 8 < std::max<int64_t>(ncols, 1);
}
// enqueues the whole product on the walked structure's stream; x entries at (d_xi, d_xv): HBM, or pinned host memory for the
// x-driven kernel.  Result: out_i / out_v / d_count (HBM) and, when `host`, the landing area + sequence number.
hipStream_t spx_enqueue(dsa_mat* h, int32_t transpose, bool xdriven, const int64_t* d_xi, const double* d_xv, int64_t nx, int64_t ny, int64_t ncols,
                        int64_t* out_i, double* out_v, int64_t cap, int64_t* d_count, long long* host, unsigned long long seq) {
    dsa_mat::Spx& x = h->spx;
    Pma& P = xdriven ? (transpose ? h->row : h->col) : (transpose ? h->col : h->row);      // the structure that is walked
    hipStream_t s = P.stream;
    // acc / bm are shared by the products of both orientations: one on another stream than the last one waits for that one's emit
    if (x.res_stream && x.res_stream != s) HIPCHK(hipStreamSynchronize(x.res_stream));
    hipError_t e;
    if (xdriven) {
        e = launch_spx_accum(P.K(), P.V(), P.O(), P.capacity(), P.sems, P.col_keys, P.col_live, P.h_ctl->table_len, d_xi, d_xv, nx, x.acc, x.bm, ny, s);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("sparse-x accumulate launch: ") + hipGetErrorString(e));
        e = launch_spx_finish(x.bm, ny, x.tile_cnt, x.tile_off, x.ticket, x.acc, 1, out_i, out_v, cap, d_count, host, SPX_PIN_CELLS, seq, s);
    } else {
        ensure_xy(h, std::max<int64_t>(2 * ncols, 1), 2 * ny);
        double* d_xd = h->d_x; double* d_xf = h->d_x + ncols;
        e = launch_scatter_x(d_xi, d_xv, nx, d_xd, d_xf, ncols, s);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("scatter launch: ") + hipGetErrorString(e));
        spmv_dev(h, transpose, 0, d_xd, ncols, h->d_y, ny, s);
        spmv_dev(h, transpose, 0, d_xf, ncols, h->d_y + ny, ny, s, 1);                      // pattern pass: touched rows
        e = launch_spx_pattern_bits(h->d_y + ny, ny, x.bm, s);
        if (e != hipSuccess) fail(DSA_EHIP, std::string("pattern launch: ") + hipGetErrorString(e));
        e = launch_spx_finish(x.bm, ny, x.tile_cnt, x.tile_off, x.ticket, h->d_y, 0, out_i, out_v, cap, d_count, host, SPX_PIN_CELLS, seq, s);
    }
    if (e != hipSuccess) fail(DSA_EHIP, std::string("sparse-x finish launch: ") + hipGetErrorString(e));
    return s;
}
// a long result (more pairs than the landing area holds) starts its way down as soon as its size is known: DMA into pinned staging in
// up to 8 pieces, an event behind each; dsa_mat_spmv_sparse_fetch copies a piece to the caller's arrays (four threads for long ones)
// while the next ones are still on the wire.  (Copies between the device and the caller's PAGEABLE arrays take anything from 3 to 14 ms
// for 16 MB depending on the state of the caller's pages; one DMA + one single-threaded copy of the whole result: 1.6 ms.)
void spx_start_download(dsa_mat* h) {
    dsa_mat::Spx& x = h->spx;
    const int64_t cnt = x.res_count;
    hipStream_t s = x.res_stream;
    spx_stage(h, (size_t)cnt * 16, s);
    char* st = static_cast<char*>(x.stage);
    const size_t tot = (size_t)cnt * 8;
    const size_t step = std::max<size_t>(((tot / 4) + 4095) & ~(size_t)4095, 1u << 18);
    int np = 0;
    for (int arr = 0; arr < 2; ++arr)
        for (size_t off = 0; off < tot; off += step) { x.dl_off[np] = (size_t)arr * tot + off; x.dl_bytes[np] = std::min(step, tot - off); ++np; }
    if ((int)x.ev.size() < np) { const size_t old = x.ev.size(); x.ev.resize((size_t)np, nullptr); for (size_t q = old; q < x.ev.size(); ++q) HIPCHK(hipEventCreateWithFlags(&x.ev[q], hipEventDisableTiming)); }
    for (int q = 0; q < np; ++q) {
        const size_t off = x.dl_off[q];
        const char* src = off < tot ? (const char*)x.oi + off : (const char*)x.ov + (off - tot);
        HIPCHK(hipMemcpyAsync(st + off, src, x.dl_bytes[q], hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(x.ev[(size_t)q], s));
    }
    x.dl_np = np; x.dl_started = true;
}
}  // namespace
extern "C" {

int32_t dsa_mat_spmv_sparse_begin(dsa_mat_t* h, int32_t transpose, const int64_t* xi, const double* xv, int64_t nx, int64_t* n_out) {
    API_TRY
    mat_flush(h);
    if (!h->has_major) fail(DSA_EMODE, "matrix is in fill mode");
    if (nx < 0) fail(DSA_EARG, "negative length");
    dsa_mat::Spx& x = h->spx;
    x.res_count = -1; x.dl_started = false;
    static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
    const auto tq0 = std::chrono::steady_clock::now();
    auto tq = [&](const char* what) { if (dbg_time) fprintf(stderr, "  [spmv_sparse_begin] %s at %.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tq0).count()); };
    const int64_t ny = transpose ? h->n : h->m;
    const int64_t ncols = transpose ? h->m : h->n;
    bool in_range = nx > 0 && xi[0] >= 1;
    const bool big_x = nx >= (int64_t)1 << 16;          // (a long x: checked while its pieces are copied to the staging area, below)
    if (!big_x) { int bad = 0; for (int64_t i = 1; i < nx; ++i) bad |= xi[i] <= xi[i - 1]; if (bad) fail(DSA_EARG, "indices of x must be strictly ascending"); }
    *n_out = 0;
    if (ny <= 0 || nx <= 0) { x.res_count = 0; return DSA_OK; }
    // (column keys below 1 are legal — test/functional/sparsematrix.jl:251 — and only the x-driven kernel can address them)
    const bool xdriven = spx_xdriven(nx, ncols) || !in_range;
    Pma& P = xdriven ? (transpose ? h->row : h->col) : (transpose ? h->col : h->row);
    hipStream_t s = P.stream;
    const bool direct = xdriven && nx <= SPX_DIRECT_X;
    spx_ensure(h, ny, direct ? 0 : nx, s);
    spx_stage(h, (size_t)nx * 16, s);
    tq("validated, scratch ready");
    char* st = static_cast<char*>(x.stage);
    const int64_t* d_xi; const double* d_xv;
    if (big_x) {
        // four threads, a quarter of x each: order check, copy into the staging area, and the piece goes on the wire while the
        // others are still being copied (one pass over the caller's arrays; 117 us check + 230 us copy + 120 us DMA one after the other before)
        constexpr int NT = 4;
        std::atomic<int> bad{0}; std::atomic<int> herr{0};
        const int64_t part = (nx + NT - 1) / NT;
        auto work = [&](int t) {
            const int64_t a = (int64_t)t * part, b = std::min<int64_t>(nx, a + part);
            if (a >= b) return;
            (void)hipSetDevice(P.device);
            int bd = 0;
            for (int64_t i = std::max<int64_t>(a, 1); i < b; ++i) bd |= xi[i] <= xi[i - 1];
            if (bd) bad.store(1);
            std::memcpy(st + (size_t)a * 8, xi + a, (size_t)(b - a) * 8);
            std::memcpy(st + (size_t)(nx + a) * 8, xv + a, (size_t)(b - a) * 8);
            if (hipMemcpyAsync(x.dx + a, st + (size_t)a * 8, (size_t)(b - a) * 8, hipMemcpyHostToDevice, s) != hipSuccess) herr.store(1);
            if (hipMemcpyAsync(x.dx + nx + a, st + (size_t)(nx + a) * 8, (size_t)(b - a) * 8, hipMemcpyHostToDevice, s) != hipSuccess) herr.store(1);
        };
        std::thread th[NT - 1];
        for (int t = 1; t < NT; ++t) th[t - 1] = std::thread(work, t);
        work(0);
        for (int t = 1; t < NT; ++t) th[t - 1].join();
        if (bad.load()) { (void)hipStreamSynchronize(s); fail(DSA_EARG, "indices of x must be strictly ascending"); }
        if (herr.load()) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); fail(DSA_EHIP, "upload of x failed"); }
        d_xi = x.dx; d_xv = reinterpret_cast<const double*>(x.dx + nx);
    } else {
        std::memcpy(st, xi, (size_t)nx * 8);
        std::memcpy(st + (size_t)nx * 8, xv, (size_t)nx * 8);
        if (direct) { d_xi = reinterpret_cast<const int64_t*>(st); d_xv = reinterpret_cast<const double*>(st + (size_t)nx * 8); }
        else {
            HIPCHK(hipMemcpyAsync(x.dx, st, (size_t)nx * 16, hipMemcpyHostToDevice, s));
            d_xi = x.dx; d_xv = reinterpret_cast<const double*>(x.dx + nx);
        }
    }
    const unsigned long long seq = ++x.seq;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    tq("x staged");
    spx_enqueue(h, transpose, xdriven, d_xi, d_xv, nx, ny, ncols, x.oi, x.ov, x.out_cap, x.d_count, x.pin, seq);
    x.res_stream = s;
    tq("enqueued");
    // the count (and a short result) arrive in the landing area: poll, asking the stream now and then (a failed launch cannot hang the host)
    wait_policy_block(P);
    volatile long long* seqp = x.pin + 1;
    auto next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    while ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) {
        if (std::chrono::steady_clock::now() < next_query) continue;
        const hipError_t q = hipStreamQuery(s);
        if (q == hipErrorNotReady) { next_query = std::chrono::steady_clock::now() + std::chrono::milliseconds(2); continue; }
        if (q != hipSuccess) fail(DSA_EHIP, std::string("sparse-x product: ") + hipGetErrorString(q));
        if ((unsigned long long)__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != seq) fail(DSA_EHIP, "sparse-x product finished without publishing its result");
    }
    x.res_count = x.pin[0];
    *n_out = x.res_count;
    tq("count back");
    if (x.res_count > SPX_PIN_CELLS) spx_start_download(h);
    API_CATCH
}

int32_t dsa_mat_spmv_sparse_fetch(dsa_mat_t* h, int64_t* yi, double* yv, int64_t cap, int64_t* n_out) {
    API_TRY
    dsa_mat::Spx& x = h->spx;
    if (x.res_count < 0) fail(DSA_EMODE, "no sparse-x product to fetch (dsa_mat_spmv_sparse_begin first)");
    const int64_t cnt = x.res_count;
    *n_out = cnt;
    if (cnt > cap) fail(DSA_ECAP, "output buffers too small");
    if (cnt == 0) return DSA_OK;
    if (cnt <= SPX_PIN_CELLS) {
        std::memcpy(yi, x.pin + 8, (size_t)cnt * 8);
        std::memcpy(yv, x.pin + 8 + SPX_PIN_CELLS, (size_t)cnt * 8);
        return DSA_OK;
    }
    // a long result: DMA into pinned staging in pieces, each piece copied to the caller's arrays (by up to four threads) while the
    // next ones are still on the wire (copies between the device and the caller's PAGEABLE arrays take anything from 3 to 14 ms for
    // 16 MB depending on the state of the caller's pages; one DMA + one single-threaded copy of the whole result: 1.6 ms)
    // a freshly allocated result array of several MB is all page faults: ask for huge pages where the system grants them on request
    // (transparent_hugepage = madvise: 2 MB faults instead of 4 KB ones; advice only, nothing changes for the caller otherwise)
    if ((size_t)cnt * 8 >= ((size_t)4 << 20))
        for (void* base : {(void*)yi, (void*)yv}) {
            const uintptr_t a = ((uintptr_t)base + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)base + (size_t)cnt * 8) & ~(uintptr_t)4095;
            if (e > a) (void)madvise((void*)a, e - a, MADV_HUGEPAGE);
        }
    if (!x.dl_started) spx_start_download(h);          // (normally on the wire since _begin learnt the count)
    const size_t tot = (size_t)cnt * 8;
    const int np = x.dl_np;
    struct Piece { char* pin; char* dst; size_t bytes; };
    Piece pc[8];
    for (int q = 0; q < np; ++q) {
        const size_t off = x.dl_off[q];
        pc[q] = Piece{static_cast<char*>(x.stage) + off, (off < tot ? (char*)yi + off : (char*)yv + (off - tot)), x.dl_bytes[q]};
    }
    static const bool dbg_time = dev_env("DSA_DBG_TIME") != nullptr;
    const auto tf0 = std::chrono::steady_clock::now();
    // every piece is copied by all workers (a quarter each) as soon as its event has fired; the workers are started ONCE per fetch
    // (a thread per piece and quarter cost more than the copies)
    const int nt = tot >= ((size_t)2 << 20) ? 4 : 1;
    std::atomic<int> herr{0};
    const int dev = P_device_of(h);
    auto work = [&](int t) {
        if (t > 0) (void)hipSetDevice(dev);
        for (int q = 0; q < np; ++q) {
            if (hipEventSynchronize(x.ev[(size_t)q]) != hipSuccess) { herr.store(1); return; }
            const size_t part = ((pc[q].bytes / nt) + 63) & ~(size_t)63;
            const size_t off = (size_t)t * part;
            if (off < pc[q].bytes) std::memcpy(pc[q].dst + off, pc[q].pin + off, std::min(part, pc[q].bytes - off));
        }
    };
    std::thread th[3];
    for (int t = 1; t < nt; ++t) th[t - 1] = std::thread(work, t);
    work(0);
    for (int t = 1; t < nt; ++t) th[t - 1].join();
    if (herr.load()) { (void)hipGetLastError(); fail(DSA_EHIP, "download of the sparse-x result failed"); }
    if (dbg_time) fprintf(stderr, "  [spmv_sparse_fetch] %lld pairs in %d pieces, %d threads: %.1f us\n", (long long)cnt, np, nt,
                          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tf0).count());
    API_CATCH
}

int32_t dsa_mat_spmv_sparse(dsa_mat_t* h, int32_t transpose, const int64_t* xi, const double* xv, int64_t nx,
                            int64_t* yi, double* yv, int64_t cap, int64_t* n_out) {
    int64_t cnt = 0;
    const int32_t rc = dsa_mat_spmv_sparse_begin(h, transpose, xi, xv, nx, &cnt);
    if (rc != DSA_OK) { *n_out = 0; return rc; }
    return dsa_mat_spmv_sparse_fetch(h, yi, yv, cap, n_out);          // (DSA_ECAP: *n_out says how much; the result stays fetchable)
}

int32_t dsa_mat_spmv_sparse_dev(dsa_mat_t* h, int32_t transpose, const int64_t* d_xi, const double* d_xv, int64_t nx,
                                int64_t* d_yi, double* d_yv, int64_t cap, int64_t* d_count) {
    API_TRY
    mat_flush(h);
    if (!h->has_major) fail(DSA_EMODE, "matrix is in fill mode");
    if (nx < 0 || cap < 0) fail(DSA_EARG, "negative length");
    const int64_t ny = transpose ? h->n : h->m;
    const int64_t ncols = transpose ? h->m : h->n;
    const bool xdriven = spx_xdriven(nx, ncols);
    Pma& P = xdriven ? (transpose ? h->row : h->col) : (transpose ? h->col : h->row);
    if (ny <= 0 || nx <= 0) { HIPCHK(hipMemsetAsync(d_count, 0, sizeof(int64_t), P.stream)); return DSA_OK; }
    spx_ensure(h, ny, 0, P.stream);
    h->spx.res_count = -1; h->spx.dl_started = false;
    h->spx.res_stream = spx_enqueue(h, transpose, xdriven, d_xi, d_xv, nx, ny, ncols, d_yi, d_yv, cap, d_count, nullptr, 0ull);
    API_CATCH
}

int32_t dsa_mat_check(dsa_mat_t* h, int32_t o, int64_t* report) { API_TRY mat_flush(h); pma_check(orient(h, o), report); API_CATCH }
int32_t dsa_mat_set_stream(dsa_mat_t* h, void* s) {
    API_TRY
    mat_flush(h);
    if (!h->has_major) fail(DSA_EMODE, "matrix is in fill mode");
    for (Pma* P : {&h->col, &h->row}) {
        HIPCHK(hipStreamSynchronize(P->stream));
        if (P->own_stream) stream_put(P->stream, P->device);
        P->stream = (hipStream_t)s; P->own_stream = false;
    }
    API_CATCH
}
int32_t dsa_mat_set_wait_policy(dsa_mat_t* h, int32_t policy) {
    API_TRY
    if (policy != DSA_WAIT_SPIN && policy != DSA_WAIT_BLOCK) fail(DSA_EARG, "wait policy must be DSA_WAIT_SPIN or DSA_WAIT_BLOCK");
    h->col.wait_policy = policy; h->row.wait_policy = policy;
    API_CATCH
}
int32_t dsa_mat_sync(dsa_mat_t* h) {
    API_TRY
    mat_flush(h);
    if (h->has_major) { HIPCHK(hipStreamSynchronize(h->col.stream)); HIPCHK(hipStreamSynchronize(h->row.stream)); }
    API_CATCH
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// parity hooks: the device slot-array primitives on a caller-supplied raw slot array (dsa_dbg_raw_*), and handles
// restored from an exported layout (dsa_*_import_layout).  Test / snapshot entry points: no reference counterpart.
// ------------------------------------------------------------------------------------------------
namespace {

// a temporary PMA around a raw slot array of `len` slots (no geometry: capacity = len, any length)
void raw_load(Pma& P, const int64_t* keys, const double* vals, const uint8_t* occ, int64_t len, const int64_t* sems, int64_t nsems,
              int64_t extra_key) {
    if (len < 1) fail(DSA_EARG, "raw slot array must hold at least one slot");
    if (nsems < 0 || (sems == nullptr && nsems > 0)) fail(DSA_EARG, "bad semaphore table");
    pma_init_common(P, sems != nullptr, false);
    P.wide = !keys_fit32(keys, len) || !key_fits32(extra_key);
    P.h_ctl->capacity = len;
    ensure_capacity_alloc(P, len);
    std::vector<uint64_t> words((size_t)P.occ_words, 0ull);
    for (int64_t i = 0; i < len; ++i) if (occ[i]) words[(size_t)(i >> 6)] |= 1ull << (i & 63);
    upload_keys(P, P.keys[P.cur], keys, len);
    HIPCHK(hipMemcpyAsync(P.V(), vals, (size_t)len * sizeof(double), hipMemcpyHostToDevice, P.stream));
    HIPCHK(hipMemcpyAsync(P.O(), words.data(), words.size() * sizeof(uint64_t), hipMemcpyHostToDevice, P.stream));
    P.occ_dirty[P.cur] = (len + 63) / 64;
    if (sems != nullptr) {
        ensure_tables(P, std::max<int64_t>(nsems, 1));
        P.h_ctl->table_len = nsems;
        if (nsems > 0) HIPCHK(hipMemcpyAsync(P.sems, sems, (size_t)nsems * sizeof(int64_t), hipMemcpyHostToDevice, P.stream));
    }
    HIPCHK(hipStreamSynchronize(P.stream));
}
void raw_store(Pma& P, int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t* sems, int64_t nsems) {
    export_slots(P, keys, vals, occ, len);
    if (sems != nullptr && nsems > 0) {
        HIPCHK(hipMemcpyAsync(sems, P.sems, (size_t)nsems * sizeof(int64_t), hipMemcpyDeviceToHost, P.stream));
        HIPCHK(hipStreamSynchronize(P.stream));
    }
}
struct RawGuard { Pma P; ~RawGuard() { if (P.stream) pma_destroy(P); } };

// one primitive on the loaded array; r = {error, position, flag, found key, found value bits, cells purged}
void raw_run(Pma& P, int32_t engine, int op, int64_t key, double val, int64_t from, int64_t to, int64_t m, int64_t r[6]) {
    hipError_t e;
    if (engine == DSA_DBG_ENGINE_BLOCK)
        e = launch_dbg_raw_block(P.K(), P.V(), P.O(), P.has_sems ? P.sems : nullptr, P.capacity(), op, key, val, from, to, m, P.d_small, P.stream);
    else if (engine == DSA_DBG_ENGINE_WAVE)
        e = launch_dbg_raw_wave(P.K(), P.V(), P.O(), P.has_sems ? P.sems : nullptr, P.capacity(), op, key, val, from, to, m, P.d_small, P.stream);
    else fail(DSA_EARG, "engine must be DSA_DBG_ENGINE_BLOCK or DSA_DBG_ENGINE_WAVE for this primitive");
    if (e != hipSuccess) fail(DSA_EHIP, std::string("parity hook launch: ") + hipGetErrorString(e));
    HIPCHK(hipMemcpyAsync(P.h_small, P.d_small, 6 * sizeof(int64_t), hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    for (int i = 0; i < 6; ++i) r[i] = P.h_small[i];
}
void check_range_args(int64_t len, int64_t from, int64_t to) {
    // find() walks [from, to] and, on a miss, left of `to` down to slot 1 (src/finds.jl:50-52): both ends must address the array
    if (from < 1 || to > len || to < 0 || from > len + 1) fail(DSA_EBOUNDS, "range outside the slot array");
}

// a handle restored from an exported layout: geometry (src/pma.jl:8-24) from capacity and segment capacity
void import_slots(Pma& P, const int64_t* keys, const double* vals, const uint8_t* occ, int64_t capacity, int64_t segment_capacity) {
    auto pow2 = [](int64_t x) { return x > 0 && (x & (x - 1)) == 0; };
    if (!pow2(capacity) || !pow2(segment_capacity) || segment_capacity > capacity / 2)
        fail(DSA_EARG, "capacity and segment capacity must be powers of two with at least two segments");
    int64_t n = 0;
    for (int64_t i = 0; i < capacity; ++i) n += occ[i] ? 1 : 0;
    std::vector<int64_t> kk((size_t)capacity);
    for (int64_t i = 0; i < capacity; ++i) kk[(size_t)i] = occ[i] ? keys[i] : 0;
    P.wide = !keys_fit32(kk.data(), capacity);
    Ctl& c = *P.h_ctl;
    c.capacity = capacity; c.segment_capacity = segment_capacity; c.nb_segments = capacity / segment_capacity;
    c.height = 0; while (((int64_t)1 << c.height) < c.nb_segments) ++c.height;
    c.nb_elements = n;
    compute_bounds(P);
    ensure_capacity_alloc(P, 2 * capacity);
    std::vector<uint64_t> words((size_t)P.occ_words, 0ull);
    for (int64_t i = 0; i < capacity; ++i) if (occ[i]) words[(size_t)(i >> 6)] |= 1ull << (i & 63);
    upload_keys(P, P.keys[P.cur], kk.data(), capacity);
    HIPCHK(hipMemcpyAsync(P.V(), vals, (size_t)capacity * sizeof(double), hipMemcpyHostToDevice, P.stream));
    HIPCHK(hipMemcpyAsync(P.O(), words.data(), words.size() * sizeof(uint64_t), hipMemcpyHostToDevice, P.stream));
    P.occ_dirty[P.cur] = (capacity + 63) / 64;
    ++P.layout_epoch;
    HIPCHK(hipStreamSynchronize(P.stream));          // the staging vectors above go out of scope
}

}  // namespace

extern "C" {

int32_t dsa_dbg_raw_find(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t len, int64_t key, int64_t from, int64_t to,
                         int32_t engine, int32_t fast, int64_t* pos, int32_t* has, int64_t* fkey, double* fval) {
    API_TRY
    check_range_args(len, from, to);
    RawGuard g;
    raw_load(g.P, keys, vals, occ, len, nullptr, 0, key);
    int64_t r[6];
    raw_run(g.P, engine, fast ? DBG_FIND_FAST : DBG_FIND, key, 0.0, from, to, 0, r);
    *pos = r[1]; *has = (int32_t)r[2]; *fkey = r[3]; std::memcpy(fval, &r[4], sizeof(double));
    API_CATCH
}
int32_t dsa_dbg_raw_insert(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t key, double value, int64_t from, int64_t to,
                           int64_t* sems, int64_t nsems, int32_t engine, int32_t fast, int64_t* pos, int32_t* is_new) {
    API_TRY
    check_range_args(len, from, to);
    RawGuard g;
    raw_load(g.P, keys, vals, occ, len, sems, nsems, key);
    int64_t r[6];
    raw_run(g.P, engine, fast ? DBG_INSERT_FAST : DBG_INSERT, key, value, from, to, 0, r);
    if (r[0] != 0) fail((int32_t)r[0], err_text((int32_t)r[0]));
    *pos = r[1]; *is_new = (int32_t)r[2];
    raw_store(g.P, keys, vals, occ, len, sems, nsems);
    API_CATCH
}
int32_t dsa_dbg_raw_delete(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t key, int64_t from, int64_t to,
                           int32_t engine, int32_t fast, int64_t* pos, int32_t* deleted) {
    API_TRY
    check_range_args(len, from, to);
    RawGuard g;
    raw_load(g.P, keys, vals, occ, len, nullptr, 0, key);
    int64_t r[6];
    raw_run(g.P, engine, fast ? DBG_DELETE_FAST : DBG_DELETE, key, 0.0, from, to, 0, r);
    *pos = r[1]; *deleted = (int32_t)r[2];
    raw_store(g.P, keys, vals, occ, len, nullptr, 0);
    API_CATCH
}
int32_t dsa_dbg_raw_purge(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t from, int64_t to, int64_t* mid, int64_t* nb) {
    API_TRY
    if (to >= from) check_range_args(len, from, to);
    RawGuard g;
    raw_load(g.P, keys, vals, occ, len, nullptr, 0, 0);
    int64_t r[6];
    raw_run(g.P, DSA_DBG_ENGINE_BLOCK, DBG_PURGE, 0, 0.0, from, to, 0, r);
    *mid = r[1]; *nb = r[5];
    raw_store(g.P, keys, vals, occ, len, nullptr, 0);
    API_CATCH
}
int32_t dsa_dbg_raw_rebalance(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t ws, int64_t we, int64_t* sems, int64_t nsems,
                              int32_t engine) {
    API_TRY
    if (ws < 1 || we > len || we < ws) fail(DSA_EBOUNDS, "window outside the slot array");
    const int64_t W = we - ws + 1;
    const bool in_word = ((ws - 1) >> 6) == ((we - 1) >> 6);
    const bool aligned = ((ws - 1) & 63) == 0 && (W & 63) == 0;
    int64_t m = 0;
    for (int64_t i = ws - 1; i < we; ++i) m += occ[i] ? 1 : 0;
    RawGuard g;
    raw_load(g.P, keys, vals, occ, len, sems, nsems, 0);
    if (engine == DSA_DBG_ENGINE_GRID) {
        // k_move2 reads whole occupancy words and writes whole destination words: windows of whole words
        if (!aligned) fail(DSA_EARG, "the grid-wide rebalance takes windows of whole occupancy words");
        if (m > 0) {
            Pma& P = g.P;
            const int alt = 1 - P.cur;
            hipError_t e = launch_rebalance(P.K(), P.V(), P.O(), ws, we, false, P.KA(alt), P.vals[alt], P.occ[alt], ws, we, m,
                                            P.has_sems ? P.sems : nullptr, &P.work, P.stream);
            if (e != hipSuccess) fail(DSA_EHIP, std::string("rebalance launch: ") + hipGetErrorString(e));
            HIPCHK(hipMemcpyAsync((char*)P.keys[P.cur] + (size_t)(ws - 1) * P.kb(), (char*)P.keys[alt] + (size_t)(ws - 1) * P.kb(), (size_t)W * P.kb(), hipMemcpyDeviceToDevice, P.stream));
            HIPCHK(hipMemcpyAsync(P.V() + (ws - 1), P.vals[alt] + (ws - 1), (size_t)W * sizeof(double), hipMemcpyDeviceToDevice, P.stream));
            HIPCHK(hipMemcpyAsync(P.O() + ((ws - 1) >> 6), P.occ[alt] + ((ws - 1) >> 6), (size_t)(W >> 6) * sizeof(uint64_t), hipMemcpyDeviceToDevice, P.stream));
        }
    } else {
        const int64_t maxw = engine == DSA_DBG_ENGINE_BLOCK ? 8192 : 2048;
        if (!(in_word || aligned) || W > maxw) fail(DSA_EARG, "window must lie inside one occupancy word or be made of whole words, within the engine's limit");
        int64_t r[6];
        raw_run(g.P, engine, DBG_REBALANCE, 0, 0.0, ws, we, m, r);
    }
    raw_store(g.P, keys, vals, occ, len, sems, nsems);
    API_CATCH
}

int32_t dsa_vec_import_layout(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t capacity, int64_t segment_capacity,
                              int64_t len, dsa_vec_t** out) {
    API_TRY
    auto* h = new dsa_vec();
    try {
        pma_init_common(h->P, false, false);
        import_slots(h->P, keys, vals, occ, capacity, segment_capacity);
        upload_ctl(h->P);
    } catch (...) { pma_destroy(h->P); delete h; throw; }
    h->n = len;
    *out = h;
    API_CATCH
}
int32_t dsa_pcsc_import_layout(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t capacity, int64_t segment_capacity,
                               const int64_t* semaphores, int64_t table_len, dsa_pcsc_t** out) {
    API_TRY
    if (table_len < 0) fail(DSA_EARG, "negative table length");
    auto* h = new dsa_pcsc();
    try {
        Pma& P = h->P;
        pma_init_common(P, true, false);
        ensure_tables(P, std::max<int64_t>(2 * table_len, 64));
        int64_t live = 0;
        for (int64_t i = 0; i < table_len; ++i) {
            const int64_t s = semaphores[i];
            if (s == 0) continue;
            if (s < 1 || s > capacity || !occ[s - 1] || keys[s - 1] != SEM_KEY || vals[s - 1] != (double)(i + 1))
                fail(DSA_EARG, "semaphores[id] must point at the cell (0, id)");
            ++live;
        }
        P.h_ctl->table_len = table_len; P.h_ctl->nb_partitions = live;
        import_slots(P, keys, vals, occ, capacity, segment_capacity);
        if (table_len > 0) HIPCHK(hipMemcpyAsync(P.sems, semaphores, (size_t)table_len * sizeof(int64_t), hipMemcpyHostToDevice, P.stream));
        upload_ctl(P);
    } catch (...) { pma_destroy(h->P); delete h; throw; }
    *out = h;
    API_CATCH
}

}  // extern "C"
