// csrc/sequencer.hip — the on-device write sequencer (K-find, K-shift, K-density, small-window
// K-rebalance) for gfx950.
//
// The reference has no batched insert: every setindex! depends on the slot layout left by the
// previous one (src/pma.jl:196-213, src/pcsr.jl:294-351).  A batch is therefore DEFINED as "apply the
// reference's setindex! in order", and one persistent 256-thread workgroup executes the whole batch
// on the device so that no host round trip is paid per element.  The control flow is executed
// redundantly and uniformly by all threads (same loads, same decisions); the data-moving steps are
// workgroup-parallel primitives:
//     d_find              exact emulation of the gap-tolerant bisection  src/finds.jl:29-57
//     blk_shift_right/left   _movecellstoright!/left! incl. semaphore fix-up  src/moves.jl:7-85
//     d_look_for_rebalance   density-threshold scan on the occupancy bitmap (popcounts) with the
//                            integer thresholds of Ctl  src/pma.jl:105-141, src/utils.jl:48-58
//     blk_rebalance_small    pack! + spread! of a window <= 8192 slots through LDS: ballot-style
//                            prefix popcounts compact the cells, closed-form gap offsets place them
//                            (src/moves.jl:94-171)
// Windows larger than that, _extend! / _shrink!, and table growth are handed back to the host
// ("yield"), which runs the grid-wide kernel of rebalance.hip and relaunches the sequencer.
#include "dsa_dev.h"
#include "find_dev.h"

namespace dsa {

// dev profile of the sequencer (Ctl::prof, printed under DSA_DBG_TIME): compiled in with -DDSA_PROFILE only — the shader-clock
// reads (s_memtime) cost a few percent of a sequential op
#ifdef DSA_PROFILE
#define DSA_TICK() ((int64_t)__builtin_readcyclecounter())
#else
#define DSA_TICK() ((int64_t)0)
#endif
constexpr int SEQ_BLOCK = 256;
constexpr int64_t SMALL_W = 8192;

// result of one op: 0 = finished, otherwise a SeqStatus; RERUN is OR-ed in when the op must be
// executed again from the top after the host has serviced the yield
constexpr int RERUN = 0x100;

struct Seq {
    KeyArr keys; double* vals; uint64_t* occ;
    int64_t* sems; int64_t* col_keys; uint8_t* col_live;
    Ctl* ctl;
    int64_t capacity, seg, height, nb_elements, nb_partitions, table_len, table_cap;
    int64_t stat_window_slots, stat_rebalances, stat_small;
    int64_t y_ws, y_we, y_m;
    int32_t err;
    bool tail_hint;                // the previous insert went behind the last cell of its range: try that first (append runs)
    const uint64_t* breaks;        // one bit per op of the batch (k_op_breaks): op j does not continue an append run from op j-1; may be null
    int64_t* sK; double* sV;       // LDS staging for the small-window rebalance
    uint32_t* sWordOff;            // [SMALL_W/64 + 1]
    int64_t* sRed;                 // [SEQ_BLOCK/64] block-reduce scratch
    const int64_t* lo; const int64_t* hi;   // integer density bounds per level (LDS copy of Ctl::lo / Ctl::hi)
    // pending partitions (see "deferred column-table inserts" below): table entries [n_sorted, table_len) were created
    // in this launch at the END of the tables, in arrival order; pKey / pIdx (LDS) list them sorted by key
    int64_t n_sorted;
    int n_pend;
    int64_t prof[16];
    int64_t* pKey; uint32_t* pIdx; uint32_t* pLb;      // pLb: number of SORTED keys below the pending key
};
constexpr int PEND_MAX = TABLE_PEND_MAX;
static_assert(PEND_MAX % SEQ_BLOCK == 0, "d_import_pending distributes the list over the workgroup");


// ---- workgroup-parallel primitives ------------------------------------------------------------------

// _nbcells(array, from, to), `to` excluded  src/utils.jl:48-58
__device__ int64_t blk_count(Seq& S, int64_t from, int64_t to) {
    if (from >= to) return 0;
    const int64_t lo0 = from - 1, hi0 = to - 2;
    const int64_t w0 = lo0 >> 6, w1 = hi0 >> 6;
    if (w1 - w0 < 8) {                                   // tiny range: every thread counts it itself
        int64_t c = 0;
        for (int64_t w = w0; w <= w1; ++w) c += popc64(S.occ[w] & word_range_mask(w, lo0, hi0));
        return c;
    }
    int64_t c = 0;
    for (int64_t w = w0 + threadIdx.x; w <= w1; w += SEQ_BLOCK) c += popc64(S.occ[w] & word_range_mask(w, lo0, hi0));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) S.sRed[threadIdx.x >> 6] = c;
    __syncthreads();
    int64_t tot = 0;
#pragma unroll
    for (int k = 0; k < SEQ_BLOCK / 64; ++k) tot += S.sRed[k];
    return tot;
}

// cells [a, b-1] move one slot to the right (b is empty)  _moverightloop!  src/moves.jl:16-42
__device__ void blk_shift_right(Seq& S, int64_t a, int64_t b) {
    for (int64_t hi = b - 1; hi >= a; hi -= SEQ_BLOCK) {
        const int64_t p = hi - threadIdx.x;
        const bool act = p >= a;
        int64_t k = 0; double v = 0.0;
        if (act) { k = S.keys[p - 1]; v = S.vals[p - 1]; }
        __syncthreads();
        if (act) {
            S.keys[p] = k; S.vals[p] = v;
            if (S.sems != nullptr && k == SEM_KEY) S.sems[(int64_t)v - 1] = p + 1;
        }
        __syncthreads();
    }
}
// cells [a+1, b] move one slot to the left (a is empty); the cell at b may itself be empty
// (last_occ == false)  _moveleftloop!  src/moves.jl:59-85
__device__ void blk_shift_left(Seq& S, int64_t a, int64_t b, bool last_occ) {
    for (int64_t lo = a + 1; lo <= b; lo += SEQ_BLOCK) {
        const int64_t p = lo + threadIdx.x;
        const bool act = p <= b && (p < b || last_occ);
        int64_t k = 0; double v = 0.0;
        if (act) { k = S.keys[p - 1]; v = S.vals[p - 1]; }
        __syncthreads();
        if (act) {
            S.keys[p - 2] = k; S.vals[p - 2] = v;
            if (S.sems != nullptr && k == SEM_KEY) S.sems[(int64_t)v - 1] = p - 1;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void occ_set(Seq& S, int64_t pos) { S.occ[(pos - 1) >> 6] |= 1ull << ((pos - 1) & 63); }
__device__ __forceinline__ void occ_clear(Seq& S, int64_t pos) { S.occ[(pos - 1) >> 6] &= ~(1ull << ((pos - 1) & 63)); }

__device__ __forceinline__ uint32_t seq_wave_excl_scan(uint32_t v) {
    const int lane = lane_id();
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x - v;
}

// pack! + spread! of [ws, we] (W <= SMALL_W) holding m cells  src/moves.jl:94-171
__device__ void blk_rebalance_small(Seq& S, int64_t ws, int64_t we, int64_t m) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t W = we - ws + 1;
    const int64_t lo0 = ws - 1;
    const int64_t w0 = lo0 >> 6;
    const SpreadGeom g = make_geom(W, m);
    if (W >= 64) {
        const int nwords = (int)(W >> 6);              // <= 128
        // exclusive prefix of the per-word popcounts: wave 0, 64 words per pass (any workgroup size)
        if (wv == 0) {
            uint32_t carry = 0;
            for (int g0 = 0; g0 < nwords; g0 += 64) {
                const int w = g0 + lane;
                const uint32_t pc = w < nwords ? (uint32_t)popc64(S.occ[w0 + w]) : 0u;
                const uint32_t ex = seq_wave_excl_scan(pc);
                if (w < nwords) S.sWordOff[w] = carry + ex;
                carry += __shfl(ex + pc, 63, 64);
            }
        }
        __syncthreads();
        for (int w = wv; w < nwords; w += SEQ_BLOCK / 64) {
            const uint64_t mask = S.occ[w0 + w];
            if ((mask >> lane) & 1ull) {
                const uint32_t r = S.sWordOff[w] + (uint32_t)popc64(mask & mask_lt(lane));
                const int64_t s = ((w0 + w) << 6) + lane;
                S.sK[r] = S.keys[s];
                S.sV[r] = S.vals[s];
            }
        }
        __syncthreads();
        for (int64_t base = 0; base < W; base += SEQ_BLOCK) {
            const int q = (int)base + tid + 1;         // 1-based offset; W is a multiple of 64, block covers 4 words
            bool occd = false;
            if (q <= W) {
                int rank;
                if (!slot_is_gap(g, q, &rank)) {
                    occd = true;
                    const int64_t k = S.sK[rank - 1];
                    const double v = S.sV[rank - 1];
                    S.keys[lo0 + q - 1] = k;
                    S.vals[lo0 + q - 1] = v;
                    if (S.sems != nullptr && k == SEM_KEY) S.sems[(int64_t)v - 1] = lo0 + q;
                }
            }
            const uint64_t b = __ballot(occd);
            if (lane == 0 && base + wv * 64 < W) S.occ[w0 + ((base + wv * 64) >> 6)] = b;
        }
    } else {
        // the window lives inside one occupancy word; wave 0 handles it
        const int bit0 = (int)(lo0 & 63);
        const uint64_t wmask = ((1ull << W) - 1ull) << bit0;
        const uint64_t word = S.occ[w0];
        __syncthreads();                                // everyone has read the old word
        if (wv == 0) {
            const uint64_t mask = (word & wmask) >> bit0;
            if (lane < W && ((mask >> lane) & 1ull)) {
                const int r = popc64(mask & mask_lt(lane));
                S.sK[r] = S.keys[lo0 + lane];
                S.sV[r] = S.vals[lo0 + lane];
            }
            // single wave: LDS writes above are visible to the same wave after the implicit wave sync
            __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0)
            bool occd = false;
            const int q = lane + 1;
            if (q <= W) {
                int rank;
                if (!slot_is_gap(g, q, &rank)) {
                    occd = true;
                    const int64_t k = S.sK[rank - 1];
                    const double v = S.sV[rank - 1];
                    S.keys[lo0 + q - 1] = k;
                    S.vals[lo0 + q - 1] = v;
                    if (S.sems != nullptr && k == SEM_KEY) S.sems[(int64_t)v - 1] = lo0 + q;
                }
            }
            const uint64_t b = __ballot(occd);
            if (lane == 0) S.occ[w0] = (word & ~wmask) | (b << bit0);
        }
    }
    __syncthreads();
    S.stat_small += 1;
}

// ---- the density-threshold scan  _look_for_rebalance!  src/pma.jl:105-141 -------------------------------
// returns 0 (nothing more to do) or a yield status
__device__ int d_after_count_change(Seq& S, int64_t pos) {
    int64_t prev_ws = pos, prev_we = pos - 1;
    int64_t left = 0, right = 0;
    int64_t ws = 1, we = S.capacity;
    bool accepted = false;
    int64_t h0 = 0;
    if (S.seg <= 64) {
        // the levels whose window fits the occupancy word of `pos` (windows are aligned powers of two): one load, popcounts of
        // sub-masks.  Only left + right is ever used, so the last window's count seeds `left` for the wider levels.
        const uint64_t word = S.occ[(pos - 1) >> 6];
        for (; h0 <= S.height; ++h0) {
            const int64_t W = S.seg << h0;
            if (W > 64) break;
            const int64_t ws0 = ((pos - 1) / W) * W;                          // 0-based first slot of the window
            const uint64_t mask = (W == 64 ? ~0ull : ((1ull << W) - 1ull)) << (ws0 & 63);
            const int64_t c = popc64(word & mask);
            ws = ws0 + 1; we = ws0 + W;
            left = c; right = 0;
            if (S.lo[h0] <= c && c <= S.hi[h0]) { accepted = true; break; }
            prev_ws = ws; prev_we = we;
        }
    }
    for (int64_t h = h0; !accepted && h <= S.height; ++h) {
        const int64_t W = S.seg << h;
        ws = ((pos - 1) / W) * W + 1;
        we = ws + W - 1;
        left += blk_count(S, ws, prev_ws);
        right += blk_count(S, prev_we + 1, we + 1);
        const int64_t c = left + right;
        if (S.lo[h] <= c && c <= S.hi[h]) { accepted = true; break; }
        prev_ws = ws; prev_we = we;
    }
    const int64_t count = left + right;
    if (!accepted) {
        const int64_t H = S.height;
        if (count > S.hi[H]) { S.y_m = count; return SEQ_Y_EXTEND; }
        if (count < S.lo[H] && S.height > 1) { S.y_m = count; return SEQ_Y_SHRINK; }
        ws = 1; we = S.capacity;
    }
    // _even_rebalance!  src/pma.jl:94-103 / src/pcsr.jl:88-97
    const int64_t W = we - ws + 1;
    if (W == S.seg) return 0;
    S.stat_rebalances += 1; S.stat_window_slots += W;
    if (W <= SMALL_W) { const int64_t tr0 = DSA_TICK(); blk_rebalance_small(S, ws, we, count); S.prof[12] += DSA_TICK() - tr0; S.prof[13] += 1; S.prof[14] += W; return 0; }
    S.y_ws = ws; S.y_we = we; S.y_m = count;
    return SEQ_Y_REBALANCE;
}

// ---- append runs: pack! / spread! simulated on the occupancy bitmap only ------------------------------------------------
// A run of setindex! calls with strictly ascending keys above the current last key (Coluna's column streaming, BASELINE
// config 2 batch A) always inserts behind the last cell, and the density scan / even rebalance that follows only needs
// cell COUNTS and produces cell POSITIONS — the keys and values never influence control flow.  Cells keep their relative
// order, so the final layout is a function of the final bitmap alone.  The run is therefore executed on the bitmap:
// wave 0 keeps the 4096-slot block around the tail in registers (one 64-bit word per lane, per-level counts as butterfly
// partial sums) and replays insert + _look_for_rebalance! + spread! per op without touching memory; windows wider than the
// block go through the workgroup-wide bitmap path.  The host then moves every cell exactly once (k_permute, rebalance.hip).
constexpr int64_t RUN_MIN = 64;
constexpr int64_t RUN_BLOCK = 4096;
constexpr int RUN_BLOCK_LOG2 = 12;

struct RunComm { int64_t idx, L, reb, slots, small; int32_t need, pad; };

// (spread_word_bits: dsa_dev.h)
// (spread_last_cell: dsa_dev.h)

// number of leading ops of ops[i..n) that continue an append run.
//   mode 0 (vector):           OP_VEC_SET, non-zero value, key above the previous key (pa0 for the first op)
//   mode 1 (MappedPackedCSC):  OP_MPCSC_SET, non-zero value, row >= 1, (col, row) lexicographically above the previous
//                              (col, row) ((pb0, pa0) for the first op; any column when first_any)
__device__ int64_t blk_run_length(Seq& S, const Op* ops, int64_t i, int64_t n, int mode, int64_t pa0, int64_t pb0, bool first_any) {
    if (S.breaks != nullptr) {
        // op i against the state of the array, the ops behind it from the bitmap of the batch (one grid-wide pass when the ops were
        // uploaded: k_op_breaks) — the scan of 100 k ops by this one workgroup took 0.4 ms per detection
        const Op o = ops[i];
        const bool bad0 = mode == 0 ? !(o.kind == OP_VEC_SET && o.v != 0.0 && o.a > pa0)
                                    : !(o.kind == OP_MPCSC_SET && o.v != 0.0 && o.a >= 1 && (first_any || o.b > pb0 || (o.b == pb0 && o.a > pa0)));
        if (bad0) return 0;
        const int64_t w0 = (i + 1) >> 6, w1 = (n - 1) >> 6;          // bits i+1 .. n-1 (bits >= the batch length are set)
        for (int64_t base = w0; base <= w1; base += SEQ_BLOCK) {
            const int64_t w = base + threadIdx.x;
            uint64_t x = w <= w1 ? S.breaks[w] : 0ull;
            if (w == w0) x &= ~mask_lt((int)((i + 1) & 63));
            int64_t first = x ? (w << 6) + __ffsll((unsigned long long)x) - 1 : INT64_MAX;
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) { const int64_t y = __shfl_xor(first, o2, 64); first = y < first ? y : first; }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) S.sRed[threadIdx.x >> 6] = first;
            __syncthreads();
            first = S.sRed[0];
#pragma unroll
            for (int k = 1; k < SEQ_BLOCK / 64; ++k) first = S.sRed[k] < first ? S.sRed[k] : first;
            if (first != INT64_MAX) return (first < n ? first : n) - i;
        }
        return n - i;
    }
    int64_t R = 0;
    for (int64_t base = i; base < n; base += SEQ_BLOCK) {
        const int64_t j = base + threadIdx.x;
        bool bad = false;
        if (j < n) {
            const Op o = ops[j];
            int64_t pa = pa0, pb = pb0;
            bool any = first_any;
            if (j != i) { const Op q = ops[j - 1]; pa = q.a; pb = q.b; any = false; }
            if (mode == 0) bad = !(o.kind == OP_VEC_SET && o.v != 0.0 && o.a > pa);
            else bad = !(o.kind == OP_MPCSC_SET && o.v != 0.0 && o.a >= 1 && (any || o.b > pb || (o.b == pb && o.a > pa)));
        }
        const uint64_t b = __ballot(bad);
        __syncthreads();
        if ((threadIdx.x & 63) == 0)
            S.sRed[threadIdx.x >> 6] = b ? (int64_t)((threadIdx.x & ~63) + __ffsll((unsigned long long)b) - 1) : (int64_t)SEQ_BLOCK;
        __syncthreads();
        int64_t first = SEQ_BLOCK;
#pragma unroll
        for (int k = 0; k < SEQ_BLOCK / 64; ++k) first = S.sRed[k] < first ? S.sRed[k] : first;
        const int64_t lim = (n - base) < SEQ_BLOCK ? (n - base) : (int64_t)SEQ_BLOCK;
        if (first < lim) return R + first;
        R += lim;
    }
    return R;
}

// one bit per op of a batch: set when op j does NOT continue an append run from op j-1 (the conditions of blk_run_length; bit 0 and the
// bits behind the last op are set).  Grid-wide, enqueued behind the upload of the ops.
__global__ __launch_bounds__(256) void k_op_breaks(const Op* ops, int64_t n, int mode, uint64_t* breaks) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool bad = true;
    if (j > 0 && j < n) {
        const Op o = ops[j], q = ops[j - 1];
        if (mode == 0) bad = !(o.kind == OP_VEC_SET && o.v != 0.0 && o.a > q.a);
        else bad = !(o.kind == OP_MPCSC_SET && o.v != 0.0 && o.a >= 1 && (o.b > q.b || (o.b == q.b && o.a > q.a)));
    }
    const uint64_t b = __ballot(bad);
    if ((threadIdx.x & 63) == 0) breaks[j >> 6] = b;
}
// the op array of a batch from the caller's columns (uploaded as they are): op k = (a[k], b ? b[k] : 0, v[k], kind)
__global__ __launch_bounds__(256) void k_make_ops(const int64_t* __restrict__ a, const int64_t* __restrict__ b, const double* __restrict__ v,
                                                  int32_t kind, int64_t n, Op* __restrict__ ops) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        Op o; o.a = a[k]; o.b = b != nullptr ? b[k] : 0; o.v = v[k]; o.kind = kind; o.pad = 0;
        ops[k] = o;
    }
}
hipError_t launch_make_ops(const int64_t* a, const int64_t* b, const double* v, int32_t kind, int64_t n, Op* ops, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_make_ops, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65536)), dim3(256), 0, stream, a, b, v, kind, n, ops);
    return hipGetLastError();
}
hipError_t launch_op_breaks(const Op* ops, int64_t n, int mode, uint64_t* breaks, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_op_breaks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ops, n, mode, breaks);
    return hipGetLastError();
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t y = __shfl_xor(v, o, 64); v = y > v ? y : v; }
    return v;
}

__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, int l) {
    return ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32) | (uint64_t)rdlane((uint32_t)v, l);
}

// memo of spread! patterns for windows of up to 256 slots: the pattern depends on (W, c) only and an append run keeps
// hitting the same few (level, count) pairs.  Filled on first use by wave 0.
// ---- model v2 of the append replay --------------------------------------------------------------------
// While the tail and the nearest gap stay inside the LAST occupancy word, an append is a handful of scalar bit operations on
// that word — set the bit behind the tail, or the nearest zero bit left of the last slot (the shift-left of _insert!,
// src/writes.jl:26-43) — and the density scan of the levels whose window fits the word (src/pma.jl:105-141) is a popcount
// of an aligned sub-mask per level; their spread! is a memoised bit pattern.  Levels wider than a word are suffixes of
// the array there: lane <-> level keeps their cell counts (+1 per append), a rebalance of level h with c cells resets the
// counts below it in closed form and yields the new last word.  The bitmap itself is written by the caller when the model
// is left: the last rebalance of every wide level that no wider one followed, widest first, then the last word.
// A separate, never inlined function with an LDS mailbox: every loop-carried value is re-established as wave-uniform
// (readfirstlane) so the loop compiles to scalar code with its own register allocation — inside the caller the same loop
// was placed in vector registers under exec masks and ran at 700 ns per op instead of ~100.
struct Model2IO {
    uint64_t lw; int64_t idx, end, reb, slots; int32_t need, progressed, nlow, seg;
    int32_t dbg_mid, dbg_miss, dbg_why, dbg_pad;      // dev counters (DSA_DBG_RUN)
    int64_t dbg_t[4];
    uint32_t cnt[64], W[64], Wb[64], lo[64], hi[64], ev_c[64], ev_valid[64];
    int32_t wb[64];
    uint32_t outside[64];        // level h wider than the block: cells of its suffix IN FRONT of the last block (workgroup pass)
    int32_t outside_valid;       // ... computed for the current bitmap
    int32_t pending;             // exit with rebalances wider than the block still to be written (workgroup), then the in-block ones
};
__device__ __forceinline__ int u32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t u64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
#ifdef DSA_M2_PROF
#define M2_T() clock64()
#else
#define M2_T() 0ll
#endif
struct RunMemo;
template <int NLOW, bool TYPED> __device__ __noinline__ void wave_model2_t(Model2IO* io, RunMemo* memo, const uint64_t* flags);

constexpr int MEMO_ENTRIES = 640, MEMO_WORDS = 1664;
// memo entries: [0, M2_DIRECT) indexed by (level, count) for windows up to 256 slots, the rest hashed by (level, count).  The memo
// lives in dynamic LDS (128 KB of the CU's 160 KB: the kernel is one workgroup)
constexpr int M2_DIRECT = 512, M2_HASHED_LOG2 = 10, M2_HASHED = 1 << M2_HASHED_LOG2, M2_EV = M2_DIRECT + M2_HASHED;
constexpr int M2_HASH_LOG2 = 11, M2_HASH = 1 << M2_HASH_LOG2;
struct RunMemo {
    uint64_t words[MEMO_WORDS]; unsigned long long gapw[64];
    // model v2: memo of the rebalances of the wide levels, tagged (level << 16 | cell count) — windows up to 256 slots have their own
    // entry (level base + count), wider ones (up to the 4096-slot block) share a direct-mapped hash: counts of the wide levels below
    // (12 bits each), the last word after the rebalance, and the EPOCH that follows it — last word, number of in-word ops, their
    // rebalances and window slots — up to the next op that needs a wide level.  Epoch word: [7:0] ops + 1 (0: none) [8] partial
    // [16:9] rebalances [30:17] window slots [63:32] tag of the entry (0xffffffff: empty).
    struct alignas(16) M2Entry { uint64_t cnt, lw, eplw, epr; } m2e[M2_EV];      // one 32-byte entry: two 16-byte LDS loads, one wait
    // epochs keyed by the last word after the rebalance and the position of the first semaphore among the ops that follow (8: none
    // among the first eight): what the entry above cannot hold — rebalances of levels without an entry, and, in a typed run, epochs
    // that run through semaphore cells (`pat`: the cell types of the epoch's ops, bit k = op k is a semaphore).  Direct-mapped.
    struct alignas(16) M2Hash { uint64_t key, eplw, epr, pat; } m2h[M2_HASH];
    Model2IO m2;             // model v2: mailbox between wave_fast_appends and wave_model2
};   // gapw: scratch of the cooperative spread

__device__ __forceinline__ int ep_n(uint64_t r) { return (int)(r & 0xffu) - 1; }
__device__ __forceinline__ int ep_reb(uint64_t r) { return (int)((r >> 9) & 0xffu); }
__device__ __forceinline__ int ep_slots(uint64_t r) { return (int)((r >> 17) & 0x3fffu); }

// NLOW = number of levels whose window fits one occupancy word, a template parameter: the segment size (64 >> (NLOW - 1)), the
// level masks and the trip count of the in-word scan are constants of each instantiation, and the levels that do not exist cost no
// scalar registers (with six run-time levels the thresholds lived in spilled SGPRs, read back with v_readlane at every use).
// TYPED = the run has cell types (MappedPackedCSC: some cells are semaphores); a vector run carries none of that code.
template <int NLOW, bool TYPED> __device__ __noinline__ void wave_model2_t(Model2IO* io_, RunMemo* memo_, const uint64_t* flags_) {
    constexpr uint64_t TOP = 1ull << 63;
    // arguments of a non-kernel function arrive in vector registers and count as divergent: re-establish them as uniform
    Model2IO* io = (Model2IO*)u64((uint64_t)io_);
    RunMemo* memo = (RunMemo*)u64((uint64_t)memo_);
    const int lane = lane_id();
    uint64_t lw = u64(io->lw);
    // ops are counted in 32 bits from a 64-aligned base (64-bit compares have no scalar form): op j is bit (j & 63) of cell-type word
    // fwp[j >> 6]; a longer run leaves at jend and comes back
    const int64_t idx0 = (int64_t)u64((uint64_t)io->idx), end0 = (int64_t)u64((uint64_t)io->end);
    const int64_t base = idx0 & ~63ll;
    const uint64_t* flags = (const uint64_t*)u64((uint64_t)flags_);
    const uint64_t* fwp = TYPED ? flags + (base >> 6) : nullptr;
    constexpr bool typed = TYPED;                    // MappedPackedCSC run: some cells are semaphores
    int j = u32((int)(idx0 - base));
    const int jend = u32((int)(end0 - base < 0x7fff0000ll ? end0 - base : 0x7fff0000ll));
    constexpr int nlow = NLOW, seg = 64 >> (NLOW - 1);
    int lo_s[NLOW], hi_s[NLOW], wb_s[NLOW];
#pragma unroll
    for (int h = 0; h < NLOW; ++h) { lo_s[h] = u32((int)io->lo[h]); hi_s[h] = u32((int)io->hi[h]); wb_s[h] = u32(io->wb[h]); }
    uint32_t cnt = io->cnt[lane];
    const uint32_t my_W = io->W[lane], my_lo = io->lo[lane], my_hi = io->hi[lane];
    const bool lvl_mid = my_W != 0;
    // memo of the wide levels: lane h holds the first entry of level h for windows up to 256 slots (entry = base + cell count);
    // wider levels hash (level, count) into [M2_DIRECT, M2_EV)
    int my_eb = -1;
    {
        int eb = 0;
        for (int h = nlow; h < 64; ++h) {
            const int Wh = seg << h;
            if (Wh > 256 || eb + Wh + 1 > M2_DIRECT) break;
            if (h == lane) my_eb = eb;
            eb += Wh + 1;
        }
    }
    auto entry_of = [&](int h, int c, int eb) -> int {
        // (levels whose lower wide levels would not fit the 5 x 12-bit counts of an entry — windows of more than 4096 slots — have none)
        return eb >= 0 ? eb + c : (h - nlow <= 5 ? M2_DIRECT + (int)((((uint32_t)h * 0x9E3779B1u) ^ ((uint32_t)c * 0x85EBCA6Bu)) >> (32 - M2_HASHED_LOG2)) : -1);
    };
    auto hslot_of = [&](uint64_t w, int p) -> int { return (int)(((w + (uint64_t)p) * 0x9E3779B97F4A7C15ull) >> (64 - M2_HASH_LOG2)); };
    const int my_k = lane - nlow < 0 ? 0 : (lane - nlow > 4 ? 4 : lane - nlow);
    const int my_sh = 12 * my_k;                     // where this level's count sits in a memo entry (at most 5 wide levels below the widest)
    uint32_t ev_c = 0;
    bool ev_valid = false;
    int reb = 0;
    int64_t slots = 0;
    int need = 0, progressed = 0, dbg_mid = 0, dbg_miss = 0, dbg_why = 0, dbg_jump = 0, dbg_ncmp = 0, dbg_sim = 0;
    int64_t dbg_tcmp = 0;
    [[maybe_unused]] int64_t pt_fast = 0, pt_sim = 0, pt_gen = 0;      // dev profile, -DDSA_M2_PROF
    [[maybe_unused]] int pn_fast = 0;
    const int64_t dbg_t0 = clock64();
    int fw_j = -1;                                   // cell-type word held in fw (index j >> 6)
    uint64_t fw = 0;
    // epoch being recorded: the in-word ops behind the rebalance that left the last word ep_key, up to the next op that needs a wide
    // level.  Cells only: it goes to the rebalance's memo entry (ep_entry >= 0) or, without one, to the hash under (ep_key, 8).  In a
    // typed run the cells in front of the first semaphore are recorded there as a PARTIAL epoch (it says nothing about the op behind
    // it), and the whole epoch with its cell types (at most 8 ops) goes to the hash under (ep_key, position of that semaphore).
    bool ep_on = false, ep_sem = false;
    int ep_entry = -1, ep_reb0 = 0, ep_j0 = 0;
    uint32_t ep_tag = 0, ep_pat = 0;
    uint64_t ep_key = 0;
    int64_t ep_slots0 = 0;
    bool wide_next = false;                          // the op at j is known to need a wide level (an epoch jump ended in front of it)
    auto pack_epoch = [&](bool partial, uint64_t hi32) -> uint64_t {      // 0: does not fit the fields
        const int n = j - ep_j0, nr = reb - ep_reb0;
        const int64_t ns = slots - ep_slots0;
        if (n > 254 || nr > 255 || ns > 16383) return 0ull;
        return (hi32 << 32) | ((uint64_t)ns << 17) | ((uint64_t)nr << 9) | (partial ? 0x100ull : 0ull) | (uint64_t)(n + 1);
    };
    auto record_plain = [&](bool partial) {          // cells only, up to (not including) op j
        const uint64_t r = pack_epoch(partial, ep_entry >= 0 ? (uint64_t)ep_tag : 0ull);
        if (r == 0 || lane != 0) return;
        if (ep_entry >= 0) { memo->m2e[ep_entry].eplw = lw; memo->m2e[ep_entry].epr = r; }
        else { RunMemo::M2Hash& hs = memo->m2h[hslot_of(ep_key, 8)]; hs.key = ep_key; hs.eplw = lw; hs.epr = r & 0xffffffffull; hs.pat = 0ull; }
    };
    auto start_epoch = [&](int entry, uint32_t tag, uint64_t key) {
        ep_on = true; ep_sem = false; ep_entry = entry; ep_tag = tag; ep_key = key; ep_pat = 0u; ep_j0 = j; ep_reb0 = reb; ep_slots0 = slots;
    };
    while (j < jend) {
        // (re-established as wave-uniform every iteration: the loop then stays on the scalar unit)
        lw = u64(lw); j = u32(j); reb = u32(reb);
        const int64_t pt0 = M2_T();
        if (wide_next) {
            // ---- the common chain: an op that needs a wide level with a memo entry, followed by the recorded epoch of that entry.
            //      Straight-line and decided before anything is modified; every other case takes the general code below.
            const uint32_t cnt2 = cnt + (cnt < my_W ? 1u : 0u);
            const uint64_t acc = __ballot(lvl_mid && my_lo <= cnt2 && cnt2 <= my_hi);
            if (acc != 0) {
                const int h = __ffsll((unsigned long long)acc) - 1;
                const int c = (int)rdlane(cnt2, h);
                const int eb = (int)rdlane((uint32_t)my_eb, h);
                const int entry = entry_of(h, c, eb);
                if (entry >= 0) {
                    const RunMemo::M2Entry en = memo->m2e[entry];
                    const uint64_t r = u64(en.epr);
                    const int n = ep_n(r);
                    bool ok = (uint32_t)(r >> 32) == (((uint32_t)h << 16) | (uint32_t)c) && n >= 0 && j + 1 + n < jend;
                    bool next_wide = (r & 0x100u) == 0;
                    if (typed && ok) {
                        // the n ops behind the event must be cells (one word of the type flags), and the op behind them too if it is to
                        // skip the in-word path
                        const int j1 = j + 1;
                        if ((j1 >> 6) != fw_j) { fw_j = j1 >> 6; fw = u64(fwp[fw_j]); }
                        ok = ((j1 + n) >> 6) == fw_j;                              // (then n <= 63)
                        const uint64_t cells = ((1ull << (n & 63)) - 1ull) << (j1 & 63);
                        ok = ok && (fw & cells) == 0;
                        next_wide = next_wide && ((fw >> ((j1 + n) & 63)) & 1ull) == 0;
                    }
                    if (ok) {
                        const uint32_t below = (uint32_t)(en.cnt >> my_sh) & 0xfffu;
                        cnt = (lane < h && lvl_mid) ? below : cnt2;
                        if (lane == h) { ev_c = (uint32_t)c; ev_valid = true; }
                        else if (lane < h) ev_valid = false;
                        const uint32_t room = my_W - cnt;                  // cnt <= my_W
                        cnt += (uint32_t)n < room ? (uint32_t)n : room;
                        j += 1; reb += 1; slots += (int64_t)(seg << h);
                        if (r & 0x100u) {
                            // a partial epoch (it was cut by a semaphore when it was recorded): keep recording from the event on, a
                            // longer one replaces it when this visit gets further
                            start_epoch(entry, (uint32_t)(r >> 32), u64(en.lw));
                        }
                        lw = u64(en.eplw);
                        j += n; reb += ep_reb(r); slots += (int64_t)ep_slots(r);
                        progressed = 1; ++dbg_mid; ++dbg_jump;
                        wide_next = next_wide;
                        pt_fast += M2_T() - pt0; ++pn_fast;
                        continue;
                    }
                }
            }
        }
        bool is_sem = false;
        if (!wide_next) {
            if (typed) {
                if ((j >> 6) != fw_j) { fw_j = j >> 6; fw = u64(fwp[fw_j]); }
                is_sem = (fw >> (j & 63)) & 1ull;
                if (is_sem && ep_on) {
                    if (!ep_sem) record_plain(true);              // the cells so far are a valid (partial) epoch of the entry
                    ep_sem = true;
                    if (j - ep_j0 < 8) ep_pat |= 1u << (j - ep_j0); else ep_on = false;
                }
            }
            // ---- the insert, on the last word  (a branch-free form of this block — selects between the three inserts, all six
            //      levels evaluated — was measured slower: 12.6 vs 10.2 ms per 41 k appends; the cost is instructions, not branches)
            uint64_t nw = lw;
            int bip;                                   // bit of the insert position
            if (lw & TOP) {                            // tail on the last slot: the cells behind the nearest gap shift left
                const uint64_t z = ~lw;
                if (z == 0) { dbg_why = 1; break; }    // that gap is in an earlier word: general path
                nw |= 1ull << (63 - __clzll((long long)z));
                bip = 63;
            } else if (!is_sem) {                      // behind the tail
                if (lw == 0) { dbg_why = 2; break; }
                bip = 64 - __clzll((long long)lw);
                nw |= 1ull << bip;
            } else {                                   // a semaphore goes to the last slot (src/pcsr.jl:99-112)
                const uint64_t z = ~lw & ~TOP;
                if (z == 0) { dbg_why = 1; break; }
                const int pe = 63 - __clzll((long long)z);
                if (pe == 62) nw |= TOP;
                else nw = (nw | (1ull << pe) | TOP) & ~(1ull << 62);
                bip = 63;
            }
            // ---- _look_for_rebalance!, levels inside the word
            int h_acc = -1, c_acc = 0, sh_acc = 0, wbh = 0;
#pragma unroll
            for (int h = 0; h < NLOW; ++h) {
                const int W = seg << h;
                const int sh = bip & ~(W - 1);
                const uint64_t m = (W == 64 ? ~0ull : ((1ull << W) - 1ull)) << sh;
                const int c = popc64(nw & m);
                if (lo_s[h] <= c && c <= hi_s[h]) { h_acc = h; c_acc = c; sh_acc = sh; wbh = wb_s[h]; break; }
            }
            if (h_acc >= 0) {
                if (h_acc > 0) {                       // _even_rebalance! inside the word: memoised pattern of (W, c)
                    const int W = seg << h_acc;
                    uint64_t pat = u64(memo->words[wbh + c_acc]);
                    if (pat == 0) {
                        ++dbg_miss;
                        SpreadGeom g;
                        g.W = W; g.E = W - c_acc;
                        g.f = (double)W / (double)(W - c_acc);
                        g.inv_f = (double)(W - c_acc) / (double)W;
                        int rank;
                        const bool cell = lane < W && !slot_is_gap(g, lane + 1, &rank);
                        pat = __ballot(cell);
                        if (lane == 0) memo->words[wbh + c_acc] = pat;
                    }
                    const uint64_t m = (W == 64 ? ~0ull : ((1ull << W) - 1ull)) << sh_acc;
                    nw = (nw & ~m) | (pat << sh_acc);
                    reb += 1; slots += W;
                }
                lw = nw;
                cnt += cnt < my_W ? 1u : 0u;
                ++j; progressed = 1; ++dbg_sim;
                pt_sim += M2_T() - pt0;
                continue;
            }
        }
        wide_next = false;
        // ---- the op needs a level wider than a word.  The epoch being recorded ends in front of it.
        if (ep_on) {
            if (!ep_sem) record_plain(false);
            else if (j - ep_j0 < 8) {
                // the whole epoch with its cell types, under (last word behind the rebalance, position of the first semaphore)
                const uint32_t pat = ep_pat | (is_sem ? 1u << (j - ep_j0) : 0u);          // bit n: the op that did not fit
                const uint64_t r = pack_epoch(false, 0ull);
                if (r != 0 && lane == 0) {
                    RunMemo::M2Hash& hs = memo->m2h[hslot_of(ep_key, __ffs((int)ep_pat) - 1)];
                    hs.key = ep_key; hs.eplw = lw; hs.epr = r; hs.pat = (uint64_t)pat;
                }
            }
            ep_on = false;
        }
        const uint32_t cnt2 = cnt + (cnt < my_W ? 1u : 0u);
        const uint64_t acc = __ballot(lvl_mid && my_lo <= cnt2 && cnt2 <= my_hi);
        if (acc == 0) { need = 1; dbg_why = 3; break; }         // a window wider than the block (or _extend!) decides
        cnt = cnt2;
        ++j; progressed = 1; ++dbg_mid;
        const int h = __ffsll((unsigned long long)acc) - 1;
        const int W = seg << h;
        const int c = (int)rdlane(cnt, h);
        reb += 1; slots += W;
        if (lane == h) { ev_c = (uint32_t)c; ev_valid = true; }
        else if (lane < h) ev_valid = false;
        const int eb = (int)rdlane((uint32_t)my_eb, h);
        const int entry = entry_of(h, c, eb);
        const uint32_t tagv = ((uint32_t)h << 16) | (uint32_t)c;
        uint64_t e = 0, en_lw = 0, en_eplw = 0, en_epr = 0;
        bool filled = false;
        if (entry >= 0) {
            const RunMemo::M2Entry en = memo->m2e[entry];
            e = u64(en.cnt); en_lw = u64(en.lw); en_eplw = u64(en.eplw); en_epr = u64(en.epr);
            filled = (uint32_t)(en_epr >> 32) == tagv;
        }
        if (!filled) {
            en_epr = 0;
            const int64_t tc0 = clock64();
            ++dbg_ncmp;
            SpreadGeom g;
            g.W = W; g.E = W - c;
            g.f = (double)W / (double)(W - c);
            g.inv_f = (double)(W - c) / (double)W;
            uint32_t below = 0;
            if (lane < h && lvl_mid) below = my_W - (uint32_t)((W - c) - gaps_le(g, W - (int)my_W));
            int rank;
            const bool cell = !slot_is_gap(g, W - 63 + lane, &rank);
            lw = __ballot(cell);
            if (lane < h && lvl_mid) cnt = below;
            if (entry >= 0) {          // counts of the (at most five) wide levels below h, 12 bits each (a level below h holds < 4096 slots)
                e = (uint64_t)rdlane(below, nlow) | ((uint64_t)rdlane(below, nlow + 1) << 12) | ((uint64_t)rdlane(below, nlow + 2) << 24) |
                    ((uint64_t)rdlane(below, nlow + 3) << 36) | ((uint64_t)rdlane(below, nlow + 4) << 48);
                if (lane == 0) { memo->m2e[entry].cnt = e; memo->m2e[entry].lw = lw; memo->m2e[entry].epr = (uint64_t)tagv << 32; }
            }
            dbg_tcmp += clock64() - tc0;
        } else {
            if (lane < h && lvl_mid) cnt = (uint32_t)(e >> my_sh) & 0xfffu;
            lw = en_lw;
        }
        // ---- the in-word ops that follow are a function of the new last word and of their cell types alone: replay them from the memo
        uint32_t pat8 = 0;                             // cell types of the next (up to 8) ops, as far as the current type word reaches
        int pat_n = 8;
        if (typed) {
            if ((j >> 6) != fw_j) { fw_j = j >> 6; fw = u64(fwp[fw_j]); }
            pat8 = (uint32_t)(fw >> (j & 63)) & 0xffu;
            pat_n = 64 - (j & 63) < 8 ? 64 - (j & 63) : 8;
            pat8 &= (1u << pat_n) - 1u;
        }
        const int p_sem = pat8 != 0 ? __ffs((int)pat8) - 1 : 8;      // position of the first semaphore among the next ops (8: none)
        // the hash: (last word, position of the first semaphore) — rebalances without an entry, and epochs that run through semaphores
        auto try_hash = [&](int p) -> bool {
            const RunMemo::M2Hash hs = memo->m2h[hslot_of(lw, p)];
            const uint64_t r = u64(hs.epr);
            const int n = ep_n(r);
            const uint32_t hp = (uint32_t)u64(hs.pat);
            bool ok = u64(hs.key) == lw && n >= 0 && j + n < jend;
            if (ok && typed) {
                if (p < 8) ok = n + 1 <= pat_n && ((pat8 ^ hp) & ((2u << n) - 1u)) == 0;          // the same types for the n ops and the op behind them
                else ok = hp == 0 && (j >> 6) == ((j + n) >> 6) && (fw & (((2ull << (n & 63)) - 1ull) << (j & 63))) == 0;
            }
            if (!ok) return false;
            lw = u64(hs.eplw);
            j += n;
            const uint32_t room = my_W - cnt;                      // cnt <= my_W
            cnt += (uint32_t)n < room ? (uint32_t)n : room;
            reb += ep_reb(r); slots += (int64_t)ep_slots(r);
            // the op behind it had this type when the epoch was recorded and needed a wide level: it does again (kept to cells)
            wide_next = (r & 0x100u) == 0 && !((hp >> n) & 1u);      // (a partial epoch says nothing about the op behind it)
            ++dbg_jump;
            if (p == 8 && entry >= 0 && lane == 0) { memo->m2e[entry].eplw = lw; memo->m2e[entry].epr = ((uint64_t)tagv << 32) | (r & 0xffffffffull); }
            return true;
        };
        bool jumped = false;
        if (p_sem < 8) jumped = try_hash(p_sem);
        if (!jumped) {
            // the cells-only epoch of the entry
            const uint64_t r = en_epr;
            const int n = ep_n(r);
            bool cells_only = true;
            if (typed && n > 0) {
                if ((j >> 6) != ((j + n - 1) >> 6)) cells_only = false;
                else cells_only = (fw & (((1ull << (n & 63)) - 1ull) << (j & 63))) == 0;
            }
            if (n >= 0 && j + n < jend && cells_only) {
                if (r & 0x100u) start_epoch(entry, tagv, lw);      // partial epoch: go on recording, a longer one replaces it
                lw = en_eplw;
                j += n;
                const uint32_t room = my_W - cnt;                  // cnt <= my_W
                cnt += (uint32_t)n < room ? (uint32_t)n : room;
                reb += ep_reb(r); slots += (int64_t)ep_slots(r);
                // the op behind a complete epoch was a cell when it was recorded (it did not fit the word): the same holds now if it is a
                // cell again; a semaphore there may still fit the word and takes the in-word path first.  A partial epoch says nothing.
                wide_next = (r & 0x100u) == 0 && (!typed || ((j >> 6) == fw_j && ((fw >> (j & 63)) & 1ull) == 0));
                ++dbg_jump;
                jumped = true;
            }
        }
        if (!jumped && p_sem == 8) jumped = try_hash(8);
        if (!jumped) start_epoch(entry, tagv, lw);
        pt_gen += M2_T() - pt0;
    }
    io->ev_c[lane] = ev_c; io->ev_valid[lane] = ev_valid ? 1u : 0u;
    if (lane == 0) { io->lw = lw; io->idx = base + j; io->need = need; io->progressed = progressed; io->reb = reb; io->slots = slots;
                     io->dbg_mid = dbg_mid; io->dbg_miss = dbg_miss; io->dbg_why = dbg_why; io->dbg_pad = dbg_sim; io->dbg_t[0] = dbg_jump; io->dbg_t[1] = dbg_ncmp; io->dbg_t[2] = dbg_tcmp; io->dbg_t[3] = clock64() - dbg_t0;
#ifdef DSA_M2_PROF
                     printf("m2 prof: fast %d events %lld clk, sim %d ops %lld clk, general %d events %lld clk, total %lld\n", pn_fast, (long long)pt_fast, dbg_sim, (long long)pt_sim, dbg_mid - pn_fast, (long long)pt_gen, (long long)(clock64() - dbg_t0));
#endif
    }
}

// wave 0: replays appends rc->idx .. end-1 on the register-resident block until one needs the workgroup path (need = 1).
// Everything per op is wave-uniform register work: lane <-> occupancy word of the block for the bitmap, lane <-> level
// for the density scan (the levels inside one word are tested first, from the word alone; wider in-block levels from a
// butterfly of word popcounts), lane <-> window offset when a spread! pattern is computed (one ballot per word).
// spread! of c cells over the W-slot window starting at slot ws, bits only, on the register-resident block (lane <-> occupancy
// word; lw0 = lane of the window's first word; wb_level = memo base of the level, used when W <= 256).  Returns the position of
// the window's last cell.
template <typename P>
__device__ __forceinline__ P wave_spread_bits(uint64_t& word, RunMemo* memo, int lane, int W, int c, int wb_level, P ws, int lw0) {
    if (W <= 256) {
        const int nw = W < 64 ? 1 : (W >> 6);
        const int wb = wb_level + c * nw;
        // a filled entry never ends with an empty word (c >= lo[h] cells spread evenly): 0 = not computed yet — one LDS
        // round trip instead of a separate valid flag
        uint64_t lastw = memo->words[wb + nw - 1];
        if (lastw == 0) {
            SpreadGeom g;                                             // make_geom(W, c) with 32-bit conversions
            g.W = W; g.E = W - c;
            g.f = (double)W / (double)(W - c);
            g.inv_f = (double)(W - c) / (double)W;
            for (int t = 0; t < nw; ++t) {
                const int q = 64 * t + lane + 1;
                int rank;
                const bool cell = q <= W && !slot_is_gap(g, q, &rank);
                const uint64_t nb = __ballot(cell);
                if (lane == 0) memo->words[wb + t] = nb;
                lastw = nb;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);                        // the other lanes read the entry below
        }
        if (W < 64) {
            const int sh = (int)((ws - 1) & 63);
            const uint64_t m = ((1ull << W) - 1ull) << sh;
            if (lane == lw0) word = (word & ~m) | (lastw << sh);
        } else {
            const int t = lane - lw0;
            if (t >= 0 && t < nw) word = memo->words[wb + t];
        }
        return ws - 1 + 64 * (nw - 1) + (64 - __clzll((long long)lastw));
    }
    SpreadGeom g;
    g.W = W; g.E = W - c;
    g.f = (double)W / (double)(W - c);
    g.inv_f = (double)(W - c) / (double)W;
    const int t = lane - lw0;
    const int nww = W >> 6, E = W - c;
    if (E <= 16 * nww) {
        // few gaps per word (dense window): the whole wave generates the E gap offsets D(k) — lane <-> k — and clears
        // their bits in an LDS image of the window, instead of every lane looping over the gaps of its own word
        if (lane < nww) memo->gapw[lane] = ~0ull;
        __builtin_amdgcn_wave_barrier();
        for (int k = lane + 1; k <= E; k += 64) {
            const int d = gap_D(g, k);                             // 1-based offset in the window
            atomicAnd(&memo->gapw[(d - 1) >> 6], ~(1ull << ((d - 1) & 63)));
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): the wave's LDS atomics have landed
        if (t >= 0 && t < nww) word = memo->gapw[t];
        __builtin_amdgcn_wave_barrier();
    } else if (t >= 0 && t < nww) word = spread_word_bits(g, t);
    return ws - 1 + (P)spread_last_cell(g);
}

// P = int32_t while capacity and run length fit 30 bits (positions, op indices and word indices are then single-register values:
// a lone wave issues ~1 instruction per 4 cycles, so halving the 64-bit arithmetic is what shortens an op), int64_t otherwise.
template <typename P>
__device__ void wave_fast_appends(Seq& S, RunComm* rc, RunMemo* memo, const uint64_t* flags, int64_t end64) {
    const P end = (P)end64;
    const int lane = lane_id();
    P idx = (P)rc->idx, L = (P)rc->L;
    int64_t reb = 0, slots = 0;
    const P cap = (P)S.capacity, seg = (P)S.seg;
    const int height = (int)S.height;
    const P nwords = (cap + 63) >> 6;
    const P blkslots = cap < (P)RUN_BLOCK ? cap : (P)RUN_BLOCK;
    const int lseg = 63 - __clzll((long long)seg);
    // level h = lane: window size and integer density bounds; levels wider than the block cannot be decided here
    const bool lvl_valid = lane <= height;
    const int64_t Wl = lvl_valid ? ((int64_t)seg << lane) : 0;
    const bool lvl_in_block = lvl_valid && Wl <= blkslots;
    const bool lvl_low = lvl_in_block && Wl <= 64;
    const bool lvl_mid = lvl_in_block && Wl > 64;
    const uint32_t my_lo = lvl_in_block ? (uint32_t)S.lo[lane] : 1u;
    const uint32_t my_hi = lvl_in_block ? (uint32_t)S.hi[lane] : 0u;
    const int my_j = lvl_mid ? (lseg + lane - 6) : 0;                           // log2(W / 64)
    const uint64_t my_low_mask = lvl_low ? (Wl == 64 ? ~0ull : ((1ull << Wl) - 1ull)) : 0ull;
    const int my_low_align = lvl_low ? ~((int)Wl - 1) & 63 : 0;
    const bool any_mid = __ballot(lvl_mid) != 0;
    // memo base of level h = lane (W <= 256): first word of entry c = 0
    int my_wb = 0;
    {
        int wb = 0;
        for (int h = 0; h < 64; ++h) {
            const int64_t Wh = (int64_t)seg << h;
            if (h > height || Wh > 256) break;
            if (h == lane) my_wb = wb;
            wb += ((int)Wh + 1) * (Wh <= 64 ? 1 : (int)(Wh >> 6));
        }
    }
    const int last_lane = (int)((nwords - 1) & 63);
    const uint64_t cap_bit = 1ull << ((cap - 1) & 63);
    int need = 0;
    // model v2: the last word is simulated bit-exactly, wider levels by their suffix counts (whole 4096-slot block in front of the end: last_lane == 63)
    const int nlow = 7 - lseg;                                      // levels whose window fits one occupancy word (W = seg << h <= 64)
    const bool model2_ok = cap >= (P)RUN_BLOCK && rc->pad == 0 && nlow >= 1 && nlow <= 6 && (int)seg << (nlow - 1) == 64;
    bool skip_model = false;                                        // the model could not place the current op: one op through the general path
    bool wrote_block = true;                                        // false: the block in registers is discarded (the workgroup rewrites wider windows first)
    const P last_blk = (cap - 1) >> RUN_BLOCK_LOG2;
    // cell types of the run (MappedPackedCSC: bit set = semaphore cell of a new column); one word per 64 cells
    P fw_idx = -1;
    uint64_t fw = 0;
    while (idx < end && need == 0) {
        if (flags != nullptr && (idx >> 6) != fw_idx) { fw_idx = idx >> 6; fw = flags[fw_idx]; }
        const bool sem0 = (fw >> (idx & 63)) & 1ull;
        const P tgt = (L < cap && !sem0) ? L + 1 : cap;
        const P blk = (tgt - 1) >> RUN_BLOCK_LOG2;
        const P w = blk * 64 + lane;
        uint64_t word = w < nwords ? S.occ[w] : 0ull;
        while (idx < end) {
            if (flags != nullptr && (idx >> 6) != fw_idx) { fw_idx = idx >> 6; fw = flags[fw_idx]; }
            const bool is_sem = (fw >> (idx & 63)) & 1ull;
            // ---- count model: the tail sits on the last slot of the array.  Every append — cell or semaphore —
            //      then takes the nearest gap left of the last slot and the windows of the density scan are the SUFFIXES of the array,
            //      so the replay only needs the suffix count of every level (lane <-> level): an append adds 1 to every level
            //      whose suffix still has a gap; a rebalance of level h with c cells leaves level j < h with W_j - G_j cells, G_j =
            //      gaps of the closed-form spread pattern that fall into the last W_j offsets.  The bitmap is not touched per op:
            //      when the model is left, the LAST rebalance of every level (each still valid outside the suffix of the next lower
            //      one) is written once and the gaps consumed since — all in the last leaf — are filled from the right.
            // ---- model v2 (wave_model2 below): the last occupancy word simulated bit-exactly, wider levels by suffix counts
            if (model2_ok && !skip_model && blk == last_blk && L > cap - 64 && !memo->m2.outside_valid) { need = 3; break; }   // the workgroup recounts first
            if (model2_ok && !skip_model && blk == last_blk && L > cap - 64) {
                Model2IO* io = &memo->m2;
                {
                    uint32_t sb[7];
                    sb[0] = (uint32_t)popc64(word);
#pragma unroll
                    for (int j = 1; j < 7; ++j) sb[j] = sb[j - 1] + __shfl_xor(sb[j - 1], 1 << (j - 1), 64);
                    const uint32_t s1 = rdlane(sb[1], 63), s2 = rdlane(sb[2], 63), s3 = rdlane(sb[3], 63), s4 = rdlane(sb[4], 63),
                                   s5 = rdlane(sb[5], 63), s6 = rdlane(sb[6], 63);
                    const uint32_t c_mid = my_j == 1 ? s1 : my_j == 2 ? s2 : my_j == 3 ? s3 : my_j == 4 ? s4 : my_j == 5 ? s5 : s6;
                    // levels wider than the block: cells in front of the block (io->outside, workgroup pass) + the whole block
                    io->cnt[lane] = lvl_mid ? c_mid : (lvl_valid && !lvl_in_block ? io->outside[lane] + s6 : 0u);
                }
                io->W[lane] = lvl_valid && Wl > 64 ? (uint32_t)Wl : 0u;
                io->Wb[lane] = lvl_in_block ? (uint32_t)Wl : 0u;
                io->lo[lane] = lvl_valid ? (uint32_t)S.lo[lane] : 1u; io->hi[lane] = lvl_valid ? (uint32_t)S.hi[lane] : 0u;
                io->wb[lane] = my_wb;
                if (lane == 0) {
                    io->lw = rdlane64(word, 63); io->idx = (int64_t)idx; io->end = (int64_t)end; io->nlow = nlow; io->seg = (int)seg;
                    io->need = 0; io->progressed = 0; io->reb = 0; io->slots = 0;
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (flags != nullptr) {
                    switch (nlow) {
                        case 1: wave_model2_t<1, true>(io, memo, flags); break;
                        case 2: wave_model2_t<2, true>(io, memo, flags); break;
                        case 3: wave_model2_t<3, true>(io, memo, flags); break;
                        case 4: wave_model2_t<4, true>(io, memo, flags); break;
                        case 5: wave_model2_t<5, true>(io, memo, flags); break;
                        default: wave_model2_t<6, true>(io, memo, flags); break;
                    }
                } else {
                    switch (nlow) {
                        case 1: wave_model2_t<1, false>(io, memo, flags); break;
                        case 2: wave_model2_t<2, false>(io, memo, flags); break;
                        case 3: wave_model2_t<3, false>(io, memo, flags); break;
                        case 4: wave_model2_t<4, false>(io, memo, flags); break;
                        case 5: wave_model2_t<5, false>(io, memo, flags); break;
                        default: wave_model2_t<6, false>(io, memo, flags); break;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (lane == 0) { S.ctl->prof[8] += 1; S.ctl->prof[9] += io->idx - (int64_t)idx; S.ctl->prof[10] += io->dbg_mid; S.ctl->prof[11] += io->dbg_miss;
                                 S.ctl->prof[12 + (io->dbg_why & 3)] += 1; S.ctl->prof[3] += io->dbg_pad; S.ctl->prof[4] += io->dbg_t[0]; S.ctl->prof[5] += io->dbg_t[1]; S.ctl->prof[6] += io->dbg_t[2]; S.ctl->prof[7] += io->dbg_t[3]; }
                idx = (P)io->idx; reb += io->reb; slots += io->slots;
                const uint64_t lw = io->lw;
                fw_idx = -1;                                   // (the callee read the cell-type words itself)
                if (lw != 0) L = cap - 63 + (P)(63 - __clzll((long long)lw));
                // the surviving wide rebalances, widest first, then the last word.  Windows wider than the block are written by the
                // whole workgroup (k_append_run), which then calls wave_apply_inblock for the rest
                const uint64_t vm_all = __ballot(io->ev_valid[lane] != 0);
                const uint64_t vm_out = vm_all & __ballot(lvl_valid && !lvl_in_block);
                if (vm_out != 0) {
                    if (lane == 0) io->pending = 1;
                    need = io->need ? 1 : 2;
                    wrote_block = false;
                    break;
                }
                uint64_t vm = vm_all;
                const uint32_t ev_c = io->ev_c[lane];
                while (vm != 0) {
                    const int j = 63 - __clzll((long long)vm);
                    vm &= ~(1ull << j);
                    const int Wj = (int)seg << j;
                    const P wsj = cap - Wj + 1;
                    wave_spread_bits<P>(word, memo, lane, Wj, (int)rdlane(ev_c, j), (int)rdlane((uint32_t)my_wb, j), wsj, (int)(((wsj - 1) >> 6) & 63));
                }
                if (lane == 63) word = lw;
                if (io->need) { need = 1; break; }
                skip_model = io->progressed == 0;
                continue;
            }
            skip_model = false;
            const uint64_t word_saved = word;
            P ip;
            if (L < cap && !is_sem) {
                // _insert! behind the last cell  src/writes.jl:26-43
                ip = L + 1;
                if (((ip - 1) >> RUN_BLOCK_LOG2) != blk) break;                       // the tail moves into the next block: reload
                if (lane == (int)(((ip - 1) >> 6) & 63)) word |= 1ull << ((ip - 1) & 63);
            } else {
                // insert at the end of the array: a cell behind a tail that sits on the last slot, or a new partition's
                // semaphore (always inserted "after position capacity", src/pcsr.jl:99-112).  The nearest empty slot left
                // of the last slot takes the shift; almost always it is in the last word.
                if (((cap - 1) >> RUN_BLOCK_LOG2) != blk) break;
                const uint64_t wl = rdlane64(word, last_lane);
                uint64_t zl = ~wl & ~cap_bit;
                if (cap < 64) zl &= (1ull << cap) - 1ull;
                uint32_t best;
                if (zl) best = (uint32_t)((last_lane << 6) + 64 - __clzll((long long)zl));
                else {
                    uint64_t z = w < nwords ? ~word : 0ull;
                    if (lane == last_lane) z &= ~cap_bit;
                    const uint32_t rel = z ? (uint32_t)((lane << 6) + 64 - __clzll((long long)z)) : 0u;
                    best = wave_max_u32(rel);
                    if (best == 0) { need = 1; break; }
                }
                const P pe = blk * (P)RUN_BLOCK + (P)best;
                ip = cap;
                if (wl & cap_bit) {                                           // tail on the last slot: cells (pe, cap] shift left
                    if (lane == (int)(((pe - 1) >> 6) & 63)) word |= 1ull << ((pe - 1) & 63);
                } else if (pe == cap - 1) {                                   // last slot and its neighbour empty
                    if (lane == last_lane) word |= cap_bit;
                } else {                                                      // last slot empty, cells (pe, cap-1] shift left
                    if (lane == (int)(((pe - 1) >> 6) & 63)) word |= 1ull << ((pe - 1) & 63);
                    if (lane == (int)(((cap - 2) >> 6) & 63)) word &= ~(1ull << ((cap - 2) & 63));
                    if (lane == last_lane) word |= cap_bit;
                }
            }
            const int lane_ip = __builtin_amdgcn_readfirstlane((int)(((ip - 1) >> 6) & 63));
            const int bit_ip = (int)((ip - 1) & 63);
            // _look_for_rebalance!  src/pma.jl:105-141: lane h evaluates level h
            const uint64_t word_ip = rdlane64(word, lane_ip);
            uint32_t c_l = (uint32_t)popc64(word_ip & (my_low_mask << (bit_ip & my_low_align)));
            uint64_t acc = __ballot(lvl_low && my_lo <= c_l && c_l <= my_hi);
            if (acc == 0 && any_mid) {
                uint32_t s[7];
                s[0] = (uint32_t)popc64(word);
#pragma unroll
                for (int j = 1; j < 7; ++j) s[j] = s[j - 1] + __shfl_xor(s[j - 1], 1 << (j - 1), 64);
                const uint32_t s1 = rdlane(s[1], lane_ip), s2 = rdlane(s[2], lane_ip), s3 = rdlane(s[3], lane_ip),
                               s4 = rdlane(s[4], lane_ip), s5 = rdlane(s[5], lane_ip), s6 = rdlane(s[6], lane_ip);
                c_l = my_j == 1 ? s1 : my_j == 2 ? s2 : my_j == 3 ? s3 : my_j == 4 ? s4 : my_j == 5 ? s5 : s6;
                acc = __ballot(lvl_mid && my_lo <= c_l && c_l <= my_hi);
            }
            if (acc == 0) {
                // a wider window (or _extend!) decides: undo, hand the op to the workgroup path
                word = word_saved;
                need = 1;
                break;
            }
            const int h = __ffsll((unsigned long long)acc) - 1;
            if (blk != last_blk && memo->m2.outside_valid) { if (lane == 0) memo->m2.outside_valid = 0; }     // cells in front of the last block changed
            if (h == 0) { L = ip; ++idx; continue; }
            // _even_rebalance!: spread! of c cells over the window, bits only
            const int W = (int)seg << h;                                      // <= 4096
            const int c = (int)rdlane(c_l, h);
            reb += 1; slots += W;
            const P ws = ((ip - 1) & ~((P)W - 1)) + 1;
            const int lw0 = (int)(((ws - 1) >> 6) & 63);
            L = wave_spread_bits<P>(word, memo, lane, W, c, (int)rdlane((uint32_t)my_wb, h), ws, lw0);
            ++idx;
        }
        if (w < nwords && wrote_block) S.occ[w] = word;
        wrote_block = true;
    }
    if (lane == 0) { rc->idx = (int64_t)idx; rc->L = (int64_t)L; rc->reb = reb; rc->slots = slots; rc->small = reb; rc->need = need; }
}

// model v2, second half of an exit that had rebalances wider than the block (written by the workgroup in between): the in-block
// survivors, widest first, then the last word, on the last block of the bitmap.  Wave 0.
__device__ void wave_apply_inblock(Seq& S, RunMemo* memo) {
    const int lane = lane_id();
    Model2IO* io = &memo->m2;
    const int64_t cap = S.capacity, seg = S.seg;
    const int height = (int)S.height;
    const int64_t last_blk = (cap - 1) >> RUN_BLOCK_LOG2;
    const int64_t w = last_blk * 64 + lane;
    uint64_t word = S.occ[w];
    int my_wb = 0;                      // memo base of level h = lane (W <= 256), as in wave_fast_appends
    {
        int wb = 0;
        for (int h = 0; h < 64; ++h) {
            const int64_t Wh = seg << h;
            if (h > height || Wh > 256) break;
            if (h == lane) my_wb = wb;
            wb += ((int)Wh + 1) * (Wh <= 64 ? 1 : (int)(Wh >> 6));
        }
    }
    const bool in_block = lane <= height && (seg << lane) <= RUN_BLOCK && (seg << lane) > 64;
    uint64_t vm = __ballot(in_block && io->ev_valid[lane] != 0);
    const uint32_t ev_c = io->ev_c[lane];
    while (vm != 0) {
        const int j = 63 - __clzll((long long)vm);
        vm &= ~(1ull << j);
        const int Wj = (int)(seg << j);
        const int64_t wsj = cap - Wj + 1;
        wave_spread_bits<int64_t>(word, memo, lane, Wj, (int)rdlane(ev_c, j), (int)rdlane((uint32_t)my_wb, j), wsj, (int)(((wsj - 1) >> 6) & 63));
    }
    if (lane == 63) word = io->lw;
    S.occ[w] = word;
}

// spread! of m cells over [ws, we], occupancy words only (workgroup-wide)
__device__ void blk_rewrite_bits(Seq& S, int64_t ws, int64_t we, int64_t m) {
    const int64_t W = we - ws + 1;
    const SpreadGeom g = make_geom(W, m);
    const int64_t w0 = (ws - 1) >> 6;
    __syncthreads();
    if (W >= 64) {
        for (int64_t t = threadIdx.x; t < (W >> 6); t += SEQ_BLOCK) S.occ[w0 + t] = spread_word_bits(g, (int)t);
    } else if (threadIdx.x == 0) {
        const int sh = (int)((ws - 1) & 63);
        const uint64_t msk = ((1ull << W) - 1ull) << sh;
        S.occ[w0] = (S.occ[w0] & ~msk) | ((spread_word_bits(g, 0) << sh) & msk);
    }
    __syncthreads();
}

// one append on the global bitmap (all threads, uniform): insert behind the tail (or, for a semaphore / a tail on the
// last slot, at the end of the array) + density scan + spread! of the bits.  Returns false (bitmap unchanged) when the op
// needs _extend! / _shrink! or cannot be placed: the run ends before it.
__device__ bool blk_slow_append(Seq& S, int64_t& L, bool is_sem) {
    const int64_t cap = S.capacity;
    int64_t ip, pe = 0;
    int how;                      // 0: set ip ; 1: set pe ; 2: set cap ; 3: set pe, clear cap-1, set cap
    if (L < cap && !is_sem) { ip = L + 1; how = 0; }
    else {
        pe = d_prev_empty(S.occ, cap);
        if (pe == 0) return false;
        ip = cap;
        how = occ_test(S.occ, cap) ? 1 : (pe == cap - 1 ? 2 : 3);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (how == 0) occ_set(S, ip);
        else if (how == 1) occ_set(S, pe);
        else if (how == 2) occ_set(S, cap);
        else { occ_set(S, pe); occ_clear(S, cap - 1); occ_set(S, cap); }
    }
    __syncthreads();
    int64_t prev_ws = ip, prev_we = ip - 1, left = 0, right = 0, ws = 1, we = cap;
    bool accepted = false;
    int64_t h0 = 0;
    if (cap >= 4096) {
        // Levels up to 65536 slots in ONE pass instead of two block-wide counts (four barriers) per level: every word of the
        // aligned 65536-slot window around ip adds its popcount to the bucket of the smallest whole-word window that holds both
        // it and ip's word; the count of a level is a prefix sum over the buckets (levels inside one word: masks of ip's word).
        __shared__ unsigned int sBk[12];
        const int64_t PW = cap < 65536 ? cap : 65536;
        const int64_t wsP = ((ip - 1) & ~(PW - 1)) + 1;
        const int64_t w0 = (wsP - 1) >> 6, nw = PW >> 6, ipw = (ip - 1) >> 6;
        if (threadIdx.x < 12) sBk[threadIdx.x] = 0u;
        __syncthreads();
        for (int64_t k = threadIdx.x; k < nw; k += SEQ_BLOCK) {
            const int64_t w = w0 + k;
            const uint64_t x = (uint64_t)(w ^ ipw);
            atomicAdd(&sBk[x ? 64 - __clzll((long long)x) : 0], (unsigned int)popc64(S.occ[w]));
        }
        __syncthreads();
        const uint64_t wip = S.occ[ipw];
        const int bip = (int)((ip - 1) & 63);
        int64_t c = 0, h = 0;
        for (; h <= S.height && (S.seg << h) <= PW; ++h) {
            const int64_t W = S.seg << h;
            if (W < 64) c = popc64(wip & ((((uint64_t)1 << W) - 1ull) << (bip & ~((int)W - 1))));
            else {
                const int lw = 63 - __clzll((long long)(W >> 6));          // window of 2^lw words
                c = 0;
                for (int k = 0; k <= lw; ++k) c += sBk[k];
            }
            if (S.lo[h] <= c && c <= S.hi[h]) { accepted = true; ws = ((ip - 1) & ~(W - 1)) + 1; we = ws + W - 1; break; }
        }
        left = c; right = 0;
        h0 = h;                                   // the loop below continues above the window counted here
        prev_ws = wsP; prev_we = wsP + PW - 1;
        __syncthreads();                          // sBk is reused by the next call
    }
    for (int64_t h = h0; !accepted && h <= S.height; ++h) {
        const int64_t W = S.seg << h;
        ws = ((ip - 1) & ~(W - 1)) + 1;
        we = ws + W - 1;
        left += blk_count(S, ws, prev_ws);
        right += blk_count(S, prev_we + 1, we + 1);
        const int64_t c = left + right;
        if (S.lo[h] <= c && c <= S.hi[h]) { accepted = true; break; }
        prev_ws = ws; prev_we = we;
    }
    const int64_t count = left + right;
    if (!accepted) {
        const int64_t H = S.height;
        if (count > S.hi[H] || (count < S.lo[H] && S.height > 1)) {
            __syncthreads();
            if (threadIdx.x == 0) {
                if (how == 0) occ_clear(S, ip);
                else if (how == 1) occ_clear(S, pe);
                else if (how == 2) occ_clear(S, cap);
                else { occ_clear(S, pe); occ_set(S, cap - 1); occ_clear(S, cap); }
            }
            __syncthreads();
            return false;
        }
        ws = 1; we = cap;
    }
    S.nb_elements += 1;
    const int64_t W = we - ws + 1;
    if (W == S.seg) { L = ip; return true; }
    S.stat_rebalances += 1; S.stat_window_slots += W;
    if (W <= SMALL_W) S.stat_small += 1;
    blk_rewrite_bits(S, ws, we, count);
    L = ws - 1 + spread_last_cell(make_geom(W, count));
    return true;
}

// Run detection (inside the sequencer): number of ops of ops[i..] that form an append run, or 0.
__device__ int64_t d_detect_append_run(Seq& S, const Op* ops, int64_t i, int64_t n_avail) {
    if (S.nb_elements < 1) return 0;
    const int64_t L0 = d_prev_occupied(S.occ, S.capacity, 1);
    if (L0 < 1) return 0;
    const int64_t lk = S.keys[L0 - 1];
    const int64_t R = blk_run_length(S, ops, i, n_avail, 0, lk, 0, false);
    return R >= RUN_MIN ? R : 0;
}
// MappedPackedCSC: (col, row) ascending above the last cell of the last partition — new columns are appended partitions
// (addpartition!(pcsc) src/pcsr.jl:99-112 via setindex! src/pcsr.jl:341-351), rows go behind the last cell.
__device__ int64_t d_detect_pcsc_run(Seq& S, const Op* ops, int64_t i, int64_t n_avail) {
    const int64_t tl = S.table_len;
    int64_t pa0 = 0, pb0 = 0;
    bool any = false;
    if (tl == 0) {
        if (S.nb_elements != 0) return 0;
        any = true;
    } else {
        if (!S.col_live[tl - 1]) return 0;                 // tombstones at the end of the table: addpartition!(pcsc, prev) paths
        const int64_t sp = S.sems[tl - 1];
        if (sp == 0) return 0;
        pb0 = S.col_keys[tl - 1];
        const int64_t L0 = d_prev_occupied(S.occ, S.capacity, 1);
        if (L0 < sp) return 0;
        const int64_t lk = L0 == sp ? 0 : S.keys[L0 - 1];
        pa0 = lk > 0 ? lk : 0;
    }
    const int64_t R = blk_run_length(S, ops, i, n_avail, 1, pa0, pb0, any);
    return R >= RUN_MIN ? R : 0;
}

// Expands the ops of a MappedPackedCSC run into its cell stream (a semaphore cell (0, id) in front of the first row of every
// new column), appends the new columns to col_keys / col_live and writes one type bit per cell.  One workgroup.
__global__ __launch_bounds__(1024) void k_run_expand(const Op* ops, int64_t i0, int64_t R, const Ctl* ctl, int64_t* col_keys,
                                                     uint8_t* col_live, Op* cells, uint64_t* flags, int64_t* out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t tl0 = ctl->table_len;
    const int64_t lastcol = tl0 > 0 ? col_keys[tl0 - 1] : 0;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < R; base += 1024) {
        const int64_t j = base + tid;
        Op o; o.a = 0; o.b = 0; o.v = 0.0; o.kind = 0; o.pad = 0;
        uint32_t newc = 0;
        if (j < R) {
            o = ops[i0 + j];
            newc = (j == 0) ? ((tl0 == 0 || o.b != lastcol) ? 1u : 0u) : (o.b != ops[i0 + j - 1].b ? 1u : 0u);
        }
        const uint32_t ex = seq_wave_excl_scan(newc);
        if (lane == 63) wsum[wv] = ex + newc;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < wv; ++k) woff += wsum[k];
        const uint32_t carry = carry_s;
        const int64_t before = (int64_t)carry + woff + ex;          // new columns in front of op j
        if (j < R) {
            const int64_t c0 = j + before;
            Op cell; cell.b = 0; cell.kind = 0; cell.pad = 0;
            if (newc) {
                cell.a = SEM_KEY; cell.v = (double)(tl0 + before + 1);
                cells[c0] = cell;
                col_keys[tl0 + before] = o.b; col_live[tl0 + before] = 1;
            }
            cell.a = o.a; cell.v = o.v;
            cells[c0 + newc] = cell;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + ex + newc;
        __syncthreads();
    }
    const int64_t T = R + (int64_t)carry_s;
    __threadfence_block();
    __syncthreads();
    for (int64_t base = 0; base < T; base += 1024) {
        const int64_t c = base + tid;
        const bool sem = c < T && cells[c].a == SEM_KEY;
        const uint64_t b = __ballot(sem);
        if (lane == 0) flags[(base >> 6) + wv] = b;
    }
    if (tid == 0) { out[0] = T; out[1] = (int64_t)carry_s; }
}

hipError_t launch_run_expand(const Op* ops, int64_t i0, int64_t R, const Ctl* ctl, int64_t* col_keys, uint8_t* col_live, Op* cells,
                             uint64_t* flags, int64_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_run_expand, dim3(1), dim3(1024), 0, stream, ops, i0, R, ctl, col_keys, col_live, cells, flags, out);
    return hipGetLastError();
}

// The run itself is its own single-workgroup kernel (own register allocation: the per-op loop of wave 0 must stay
// spill-free).  Replays ops [i0, i0+R) on the bitmap `occ` (the saved copy for K-permute was made by the host) and
// updates the control block: next_op, nb_elements, statistics.  A run ends early at an op that needs _extend! /
// _shrink!; when that is the very first op, no_run_at tells the sequencer to execute it on the normal path.
__global__ __launch_bounds__(SEQ_BLOCK) void k_append_run(uint64_t* occ, Ctl* ctl, int64_t i0, int64_t R, const uint64_t* flags,
                                                          const int64_t* d_T, int wide_pos, int no_model, uint64_t* saved_memo,
                                                          const int64_t* m3_out) {
    __shared__ int64_t sRed[SEQ_BLOCK / 64];
    __shared__ RunComm sRun;
    extern __shared__ __attribute__((aligned(16))) unsigned char run_lds[];
    RunMemo& sMemo = *reinterpret_cast<RunMemo*>(run_lds);
    __shared__ int64_t sLo[MAX_LEVELS], sHi[MAX_LEVELS];
    // The memo is a pure function of the array's geometry (capacity, segment size, the integer thresholds of every level) and of
    // the kind of run: it survives from one run of the handle to the next in HBM (config 5: 50 runs of 17 000 cells on the same
    // geometry; a cold memo re-learns ~3000 (last word, cell types) epochs from the 1000 semaphores of each run) and is dropped
    // when the tag — a hash of exactly those inputs — changes (_extend!).
    uint64_t memo_tag = 0xcbf29ce484222325ull ^ (uint64_t)ctl->capacity;
    memo_tag = (memo_tag * 0x100000001b3ull) ^ (uint64_t)ctl->segment_capacity;
    memo_tag = (memo_tag * 0x100000001b3ull) ^ (uint64_t)(flags != nullptr ? 2 : 1);
    for (int h = 0; h <= (int)ctl->height && h < MAX_LEVELS; ++h) {
        memo_tag = (memo_tag * 0x100000001b3ull) ^ (uint64_t)ctl->lo[h];
        memo_tag = (memo_tag * 0x100000001b3ull) ^ (uint64_t)ctl->hi[h];
    }
    constexpr int MEMO_U4 = (int)(sizeof(RunMemo) / 16);
    static_assert(sizeof(RunMemo) % 16 == 0, "the memo is saved in 16-byte pieces");
    // the count-only replay (appendmodel.hip: k_append_model3, launched in front of this kernel) has placed the first m3_idx cells
    // on the bitmap and counted them in the control block; m3_ended: it stopped in front of an op that no level accepts
    const int64_t m3_idx = (m3_out != nullptr && m3_out[1] != 0) ? m3_out[0] : 0;
    const bool m3_ended = m3_out != nullptr && m3_out[1] == 2;
    const int64_t end_cells = flags != nullptr ? d_T[0] : R;
    const bool m3_all = m3_ended || (m3_out != nullptr && m3_out[1] != 0 && m3_idx >= end_cells);      // nothing left for this kernel's replay
    const bool memo_warm = !m3_all && saved_memo != nullptr && saved_memo[2 * MEMO_U4] == memo_tag;
    if (m3_all) {
    } else if (memo_warm) {
        const uint4* src = reinterpret_cast<const uint4*>(saved_memo);
        uint4* dst = reinterpret_cast<uint4*>(run_lds);
        for (int k = threadIdx.x; k < MEMO_U4; k += SEQ_BLOCK) dst[k] = src[k];
    } else {
        for (int k = threadIdx.x; k < MEMO_WORDS; k += SEQ_BLOCK) sMemo.words[k] = 0ull;      // 0 = entry not computed yet
        for (int k = threadIdx.x; k < M2_EV; k += SEQ_BLOCK) { sMemo.m2e[k].cnt = 0ull; sMemo.m2e[k].epr = ~0ull << 32; }
        for (int k = threadIdx.x; k < M2_HASH; k += SEQ_BLOCK) { sMemo.m2h[k].key = 0ull; sMemo.m2h[k].epr = 0ull; }
    }
    Seq S;
    S.keys = KeyArr{nullptr, 1, 0}; S.vals = nullptr; S.occ = occ; S.sems = nullptr; S.col_keys = nullptr; S.col_live = nullptr; S.ctl = ctl;
    S.capacity = ctl->capacity; S.seg = ctl->segment_capacity; S.height = ctl->height;
    S.nb_elements = ctl->nb_elements; S.nb_partitions = 0; S.table_len = 0; S.table_cap = 0;
    S.stat_window_slots = ctl->stat_window_slots; S.stat_rebalances = ctl->stat_rebalances;
    S.stat_small = ctl->stat_small_rebalances;
    S.y_ws = S.y_we = S.y_m = 0; S.err = 0; S.tail_hint = false; S.breaks = nullptr;
    S.sK = nullptr; S.sV = nullptr; S.sWordOff = nullptr; S.sRed = sRed;
    if (threadIdx.x < MAX_LEVELS) { sLo[threadIdx.x] = ctl->lo[threadIdx.x]; sHi[threadIdx.x] = ctl->hi[threadIdx.x]; }
    S.lo = sLo; S.hi = sHi;
    __syncthreads();
    RunComm* rc = &sRun;
    // cells of the run: the ops themselves (vector), or the expanded cell stream of k_run_expand (flags != nullptr)
    int64_t idx = m3_idx, L = d_prev_occupied(occ, S.capacity, 1);
    const int64_t end = end_cells;
    int64_t t_fast = 0, t_slow = 0, n_slow = 0;
    const bool narrow_pos = !wide_pos && S.capacity <= (1ll << 30) && end <= (1ll << 30);
    const int64_t c_begin = clock64(), w_begin = wall_clock64();
    const bool use_v2 = no_model == 0 && S.capacity >= RUN_BLOCK;
    if (threadIdx.x == 0) { sMemo.m2.outside_valid = 0; sMemo.m2.pending = 0; }
    __syncthreads();
    while (idx < end && !m3_ended) {
        if (use_v2 && !sMemo.m2.outside_valid) {
            // model v2: suffix cell counts of the levels wider than the block, without the last block (all threads, uniform)
            for (int j = 0; j <= (int)S.height && j < 64; ++j) {
                const int64_t W = S.seg << j;
                if (W > RUN_BLOCK && W <= S.capacity) {
                    const int64_t c = blk_count(S, S.capacity - W + 1, S.capacity - RUN_BLOCK + 1);
                    if (threadIdx.x == 0) sMemo.m2.outside[j] = (uint32_t)c;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) sMemo.m2.outside_valid = 1;
        }
        if (threadIdx.x == 0) { rc->idx = idx; rc->L = L; rc->pad = no_model; }
        __syncthreads();
        const int64_t t0 = wall_clock64();
        if (threadIdx.x < 64) {
            if (narrow_pos) wave_fast_appends<int32_t>(S, rc, &sMemo, flags, end);
            else wave_fast_appends<int64_t>(S, rc, &sMemo, flags, end);
        }
        __syncthreads();
        const int64_t t1 = wall_clock64();
        t_fast += t1 - t0;
        const int64_t nidx = rc->idx;
        L = rc->L;
        S.nb_elements += nidx - idx;
        S.stat_rebalances += rc->reb; S.stat_window_slots += rc->slots; S.stat_small += rc->small;
        idx = nidx;
        const int need = rc->need;
        __syncthreads();
        if (sMemo.m2.pending) {
            // model v2 left with rebalances wider than the block: the workgroup writes them, widest first, then wave 0 the rest
            for (int j = (int)S.height; j >= 0; --j) {
                const int64_t W = S.seg << j;
                if (W > RUN_BLOCK && W <= S.capacity && sMemo.m2.ev_valid[j] != 0) blk_rewrite_bits(S, S.capacity - W + 1, S.capacity, (int64_t)sMemo.m2.ev_c[j]);
            }
            __syncthreads();
            if (threadIdx.x < 64) wave_apply_inblock(S, &sMemo);
            __syncthreads();
            if (threadIdx.x == 0) { sMemo.m2.pending = 0; sMemo.m2.outside_valid = 0; }
            __syncthreads();
        }
        if (need != 1) continue;
        if (threadIdx.x == 0) sMemo.m2.outside_valid = 0;      // (the slow op changes the bitmap in front of the block)
        const bool ok = blk_slow_append(S, L, flags != nullptr && ((flags[idx >> 6] >> (idx & 63)) & 1ull));
        t_slow += wall_clock64() - t1; ++n_slow;
        if (!ok) break;
        ++idx;
    }
    // semaphore cells among the idx cells that were placed: new partitions; the others are completed ops
    int64_t nsem = 0;
    if (flags != nullptr) {
        for (int64_t wd = threadIdx.x; (wd << 6) < idx; wd += SEQ_BLOCK) {
            uint64_t f = flags[wd];
            if (((wd + 1) << 6) > idx) f &= mask_lt((int)(idx - (wd << 6)));
            nsem += popc64(f);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nsem += __shfl_xor(nsem, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = nsem;
        __syncthreads();
        nsem = 0;
        for (int k = 0; k < SEQ_BLOCK / 64; ++k) nsem += sRed[k];
    }
    if (saved_memo != nullptr && !m3_all) {
        __syncthreads();
        const uint4* src = reinterpret_cast<const uint4*>(run_lds);
        uint4* dst = reinterpret_cast<uint4*>(saved_memo);
        for (int k = threadIdx.x; k < MEMO_U4; k += SEQ_BLOCK) dst[k] = src[k];
        if (threadIdx.x == 0) saved_memo[2 * MEMO_U4] = memo_tag;
    }
    if (threadIdx.x == 0) {
        const int64_t next = i0 + idx - nsem;
        ctl->next_op = next;
        ctl->no_run_at = idx < end ? next : -1;
        ctl->nb_partitions += nsem; ctl->table_len += nsem;
        ctl->nb_elements = S.nb_elements;
        ctl->stat_window_slots = S.stat_window_slots; ctl->stat_rebalances = S.stat_rebalances;
        ctl->stat_small_rebalances = S.stat_small;
        ctl->dbg[0] = n_slow; ctl->dbg[2] = t_fast; ctl->dbg[3] = t_slow; ctl->dbg[4] = idx;
        ctl->dbg[1] = clock64() - c_begin; ctl->dbg[5] = wall_clock64() - w_begin;
    }
}

size_t append_run_memo_bytes() { return sizeof(RunMemo) + 16; }

hipError_t launch_append_run(uint64_t* occ, Ctl* ctl, int64_t i0, int64_t R, const uint64_t* flags, const int64_t* d_T,
                             uint64_t* saved_memo, const int64_t* m3_out, hipStream_t stream) {
    static const int wide_pos = [] { const char* e = dev_env("DSA_POS_WIDE"); return (e && e[0] == '1') ? 1 : 0; }();   // dev knob: 64-bit positions
    // dev knob DSA_COUNT_MODEL=0: bitmap replay only (the general per-op path; A/B runs and coverage of that path)
    static const int no_model = [] { const char* e = dev_env("DSA_COUNT_MODEL"); return (e && e[0] == '0') ? 1 : 0; }();
    static PerDeviceOnce once;
    {
        hipError_t e = once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(k_append_run), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RunMemo));
        });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_append_run, dim3(1), dim3(SEQ_BLOCK), sizeof(RunMemo), stream, occ, ctl, i0, R, flags, d_T, wide_pos, no_model, saved_memo, m3_out);
    return hipGetLastError();
}

// _insert!(array, key, value, pos, semaphores)  src/writes.jl:26-43 ; returns the insertion position or 0 (EFULL)
__device__ int64_t d_insert_after(Seq& S, int64_t key, double val, int64_t pos) {
    const int64_t ne = d_next_empty(S.occ, pos, S.capacity);
    const int64_t pe = ne != 0 ? 0 : d_prev_empty(S.occ, pos);
    const bool last_occ = (ne == 0 && pe != 0) ? occ_test(S.occ, pos) : true;
    if (ne != 0 && ne - (pos + 1) <= SEQ_BLOCK) {
        // common case, fused: one chunk.  bitmap reads + cell loads | barrier | cell stores + new cell + bitmap | barrier
        const int64_t p = ne - 1 - threadIdx.x;
        const bool act = p >= pos + 1;
        int64_t k = 0; double v = 0.0;
        if (act) { k = S.keys[p - 1]; v = S.vals[p - 1]; }
        __syncthreads();
        if (act) {
            S.keys[p] = k; S.vals[p] = v;
            if (S.sems != nullptr && k == SEM_KEY) S.sems[(int64_t)v - 1] = p + 1;
        }
        if (threadIdx.x == 0) { S.keys[pos] = key; S.vals[pos] = val; occ_set(S, ne); }
        __syncthreads();
        return pos + 1;
    }
    __syncthreads();                       // every thread has finished reading the bitmap
    if (ne != 0) {
        blk_shift_right(S, pos + 1, ne);
        if (threadIdx.x == 0) { S.keys[pos] = key; S.vals[pos] = val; occ_set(S, ne); }
        __syncthreads();
        return pos + 1;
    }
    if (pe == 0) { S.err = E_FULL; return 0; }
    blk_shift_left(S, pe, pos, last_occ);
    if (threadIdx.x == 0) {
        S.keys[pos - 1] = key; S.vals[pos - 1] = val;
        // occupancy after the shift: bits [pe, pos-1] take the old bits [pe+1, pos]; (pe, pos-1] were all ones
        if (pe < pos - 1) { occ_set(S, pe); if (!last_occ) occ_clear(S, pos - 1); }
        else if (last_occ) occ_set(S, pe);
        occ_set(S, pos);
    }
    __syncthreads();
    return pos;
}

// setindex!(pma, value, key)  src/pma.jl:196-213 restricted to [from, to] like insert!/delete! of
// src/writes.jl:14-23,57-63 ; del_from is the (possibly wider) range of the delete path (src/pcsr.jl:302-307)
__device__ int d_set_in_range(Seq& S, int64_t key, double val, int64_t from, int64_t to, int64_t del_from) {
    if (val != 0.0) {
        // append runs (ascending keys: Coluna's column streaming, BASELINE config 2 batch A): when the previous insert landed
        // behind the last cell of its range, look there first — find() returns that cell whenever its key is smaller
        DFound f;
        bool have = false;
        if (S.tail_hint && to >= from) {
            const int64_t lp = d_prev_occupied(S.occ, to, from);
            if (lp >= from) {
                const int64_t lk = S.keys[lp - 1];
                if (lk < key) { f = DFound{lp, lk, 0.0, true}; have = true; }
            }
        }
        const int64_t ts0 = DSA_TICK();
        if (!have) f = d_find_fast(S.keys, S.vals, S.occ, key, from, to);     // [from, to] never holds a semaphore
        S.prof[8] += DSA_TICK() - ts0;
        S.tail_hint = have || (f.has && f.pos >= from && f.key < key && d_next_occupied(S.occ, f.pos, to) == 0);
        if (f.has && f.key == key && from <= f.pos && f.pos <= to) {
            __syncthreads();
            if (threadIdx.x == 0) S.vals[f.pos - 1] = val;
            __syncthreads();
            return 0;
        }
        const int64_t ts1 = DSA_TICK();
        const int64_t ip = d_insert_after(S, key, val, f.pos);
        if (ip == 0) return SEQ_ERROR;
        S.nb_elements += 1;
        const int64_t ts2 = DSA_TICK();
        const int rr = d_after_count_change(S, ip);
        S.prof[9] += ts2 - ts1; S.prof[10] += DSA_TICK() - ts2;
        return rr;
    }
    // the delete range of a partition starts AT its semaphore (key 0, src/pcsr.jl:307): the wave-parallel search is
    // only equivalent for key > 0 there; otherwise replay the reference bisection probe for probe
    const bool sorted_ok = (S.sems == nullptr) || key > SEM_KEY;
    const DFound f = sorted_ok ? d_find_fast(S.keys, S.vals, S.occ, key, del_from, to)
                               : d_find(S.keys, S.vals, S.occ, key, del_from, to);
    if (f.has && f.key == key) {
        __syncthreads();
        if (threadIdx.x == 0) occ_clear(S, f.pos);
        __syncthreads();
        S.nb_elements -= 1;
        return d_after_count_change(S, f.pos);
    }
    return 0;
}

// addpartition!(pcsc)  src/pcsr.jl:99-112
__device__ int d_addpartition_append(Seq& S) {
    if (S.table_len + 1 > S.table_cap) return SEQ_Y_TABLE_GROW | RERUN;
    const int64_t sem_pos = S.capacity;
    S.nb_partitions += 1;
    __syncthreads();
    if (threadIdx.x == 0) S.sems[S.table_len] = sem_pos;
    __syncthreads();
    S.table_len += 1;
    const double sem_val = (double)S.table_len;
    const int64_t ip = d_insert_after(S, SEM_KEY, sem_val, sem_pos);
    if (ip == 0) return SEQ_ERROR;
    S.nb_elements += 1;
    const int r = d_after_count_change(S, ip);
    return r ? (r | RERUN) : 0;
}

// addcolumn! (src/pcsr.jl:148-169) + addpartition!(pcsc, prev) (src/pcsr.jl:114-146); col_keys may be null (plain PackedCSC)
__device__ int d_addpartition_middle(Seq& S, int64_t prev, bool with_col, int64_t col) {
    // prev = prev_sem_id = prev_col_pos ; the new partition gets id prev+1 (0-based table index prev)
    if (prev + 1 < 1 || prev + 1 > S.table_len) { S.err = E_BOUNDS; return SEQ_ERROR; }
    const int64_t target = S.sems[prev];
    int64_t sem_pos = 0;
    if (target == 0) {
        // addcolumn! stores the column key in the tombstoned slot BEFORE addpartition! looks for the next semaphore
        // (src/pcsr.jl:155-156,165): the key stays there when that lookup throws
        __syncthreads();
        if (with_col && threadIdx.x == 0) { S.col_keys[prev] = col; S.col_live[prev] = 1; }
        __syncthreads();
        const int64_t next = d_next_live_sem(S.sems, prev + 1, S.table_len);
        if (next == 0) { S.err = E_BOUNDS; return SEQ_ERROR; }      // semaphores[0] in the reference (App. A.6 (3))
        sem_pos = S.sems[next - 1] - 1;
    } else {
        if (S.table_len + 1 > S.table_cap) return SEQ_Y_TABLE_GROW | RERUN;
        // reference @assert !isnothing(moved_sem_pos) (src/pcsr.jl:132): no tombstone may be shifted
        int bad = 0;
        for (int64_t i = prev + threadIdx.x; i < S.table_len; i += SEQ_BLOCK) if (S.sems[i] == 0) bad = 1;
        if (__syncthreads_or(bad)) { S.err = E_ASSERT; return SEQ_ERROR; }
        sem_pos = target - 1;
        // shift tables one entry to the right from index prev, highest chunk first
        for (int64_t hi = S.table_len - 1; hi >= prev; hi -= SEQ_BLOCK) {
            const int64_t i = hi - threadIdx.x;
            const bool act = i >= prev;
            int64_t sp = 0, ck = 0; uint8_t lv = 0;
            if (act) { sp = S.sems[i]; if (with_col) { ck = S.col_keys[i]; lv = S.col_live[i]; } }
            __syncthreads();
            if (act) {
                S.sems[i + 1] = sp;
                if (with_col) { S.col_keys[i + 1] = ck; S.col_live[i + 1] = lv; }
                S.vals[sp - 1] = (double)(i + 2);       // the semaphore of id i+1 becomes id i+2
            }
            __syncthreads();
        }
        if (with_col && threadIdx.x == 0) { S.col_keys[prev] = col; S.col_live[prev] = 1; }
        S.table_len += 1;
    }
    __syncthreads();
    S.nb_partitions += 1;
    const double sem_val = (double)(prev + 1);
    const int64_t ip = d_insert_after(S, SEM_KEY, sem_val, sem_pos);
    if (ip == 0) return SEQ_ERROR;
    if (threadIdx.x == 0) S.sems[prev] = ip;
    __syncthreads();
    S.nb_elements += 1;
    const int r = d_after_count_change(S, ip);
    return r ? (r | RERUN) : 0;
}

// ---- deferred column-table inserts ----------------------------------------------------------------------------------
// addpartition!(pcsc, prev) in the middle of the tables (src/pcsr.jl:114-146) shifts semaphores[] / col_keys[] one entry to
// the right and rewrites the id stored in EVERY later semaphore cell: O(#partitions) per new column — the cost of streaming
// new rows into the rowmajor twin in random key order (BASELINE config 5: 100k rows -> 5e9 id rewrites).  Ids are only
// labels: the slot layout depends on WHERE the new semaphore cell is inserted (in front of the semaphore of its successor in
// key order), not on its number.  So inside one launch a new partition is appended at the END of the tables with the next
// free id (cells and tables stay consistent with each other: shifts, rebalances and spreads keep working on ids), its key
// is kept in a small sorted list in LDS, lookups consult the sorted table part + that list, and the tables are brought
// back to key order — ONE merge pass with ONE renumbering of the cells — when the list is full and before any op that is not
// a plain MappedPackedCSC write (d_merge_pending below, one workgroup), or by the host between launches and before the batch
// returns (tables.hip: the same pass with the whole chip; entries still pending when the kernel exits are counted in
// Ctl::n_pending and re-imported by the next launch; the big rebalance and the batch-parallel kernels work on ids and do not
// care about the order).  Only used while no tombstone exists (nb_partitions == table_len); the tombstone-reuse and @assert
// paths of the reference run on the literal code below.
__device__ int pend_lower_bound(const Seq& S, int64_t key) {          // first j with pKey[j] >= key
    int lo = 0, hi = S.n_pend;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (S.pKey[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ void pend_insert(Seq& S, int64_t key, int64_t idx, int64_t lb, int p) {
    __syncthreads();
    int64_t kk[PEND_MAX / SEQ_BLOCK]; uint32_t ii[PEND_MAX / SEQ_BLOCK], ll[PEND_MAX / SEQ_BLOCK];
#pragma unroll
    for (int u = 0; u < PEND_MAX / SEQ_BLOCK; ++u) {
        const int j = p + threadIdx.x + SEQ_BLOCK * u;
        if (j < S.n_pend) { kk[u] = S.pKey[j]; ii[u] = S.pIdx[j]; ll[u] = S.pLb[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PEND_MAX / SEQ_BLOCK; ++u) {
        const int j = p + threadIdx.x + SEQ_BLOCK * u;
        if (j < S.n_pend) { S.pKey[j + 1] = kk[u]; S.pIdx[j + 1] = ii[u]; S.pLb[j + 1] = ll[u]; }
    }
    if (threadIdx.x == 0) { S.pKey[p] = key; S.pIdx[p] = (uint32_t)idx; S.pLb[p] = (uint32_t)lb; }
    __syncthreads();
    S.n_pend += 1;
}
// successor in key order of `key` (which is in neither set, or is the key of partition `self`): 0-based table index, -1 if none.
// sorted_succ = 0-based index of the first sorted entry with a larger key (>= n_sorted: none)
__device__ int64_t succ_index(const Seq& S, int64_t sorted_succ, int64_t key) {
    const int pj = pend_lower_bound(S, key + 1);
    const bool has_s = sorted_succ < S.n_sorted, has_p = pj < S.n_pend;
    if (has_s && has_p) return S.col_keys[sorted_succ] < S.pKey[pj] ? sorted_succ : (int64_t)S.pIdx[pj];
    if (has_s) return sorted_succ;
    if (has_p) return (int64_t)S.pIdx[pj];
    return -1;
}
// tables back to key order: sorted entry i moves up by the number of pending keys below it, pending rank r goes to
// (#sorted keys below it) + r; every cell whose id changed is rewritten once
__device__ void d_merge_pending(Seq& S) {
    const int K = S.n_pend;
    if (K == 0) return;
    const int64_t tm0 = DSA_TICK();
    __syncthreads();
    const int64_t ns = S.n_sorted;
    int64_t* dst = S.sK;                                      // dynamic LDS is free between ops
    int64_t* spos = reinterpret_cast<int64_t*>(S.sV);
    for (int r = threadIdx.x; r < K; r += SEQ_BLOCK) {
        dst[r] = (int64_t)S.pLb[r] + r;                       // (#sorted keys below it) + (#pending keys below it)
        spos[r] = S.sems[S.pIdx[r]];
    }
    __syncthreads();
    const int64_t i_min = dst[0];                              // sorted entries below the smallest pending key stay
    constexpr int U = 4;
    for (int64_t hi = ns - 1; hi >= i_min; hi -= SEQ_BLOCK * U) {
        int64_t sp[U], ck[U]; int sh[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = hi - threadIdx.x - SEQ_BLOCK * u;
            if (i >= i_min) { sp[u] = S.sems[i]; ck[u] = S.col_keys[i]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = hi - threadIdx.x - SEQ_BLOCK * u;
            sh[u] = i >= i_min ? pend_lower_bound(S, ck[u]) : 0;
        }
        __syncthreads();               // all loads of the chunk before its stores; stores never reach below the chunk
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = hi - threadIdx.x - SEQ_BLOCK * u;
            if (i >= i_min && sh[u] > 0) {
                const int64_t d = i + sh[u];
                S.sems[d] = sp[u]; S.col_keys[d] = ck[u]; S.col_live[d] = 1;
                S.vals[sp[u] - 1] = (double)(d + 1);
            }
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < K; r += SEQ_BLOCK) {
        const int64_t d = dst[r];
        S.sems[d] = spos[r]; S.col_keys[d] = S.pKey[r]; S.col_live[d] = 1;
        S.vals[spos[r] - 1] = (double)(d + 1);
    }
    __syncthreads();
    S.n_sorted = S.table_len;
    S.n_pend = 0;
    S.prof[3] += DSA_TICK() - tm0; S.prof[6] += 1;
}

// entries left at the end of the tables by earlier launches of the batch (Ctl::n_pending, arrival order: batch-parallel rounds and
// previous sequencer chunks): rebuild the sorted list; the merge itself is the host's grid-wide pass (tables.hip) or, when the
// list fills up inside this launch, d_merge_pending
__device__ void d_import_pending(Seq& S, int64_t K) {
    if (K <= 0) return;
    const int64_t ns = S.table_len - K;
    S.n_sorted = ns;
    int64_t* tmp = S.sK;                                       // dynamic LDS is free between ops
    for (int64_t r = threadIdx.x; r < K; r += SEQ_BLOCK) tmp[r] = S.col_keys[ns + r];
    __syncthreads();
    for (int64_t r = threadIdx.x; r < K; r += SEQ_BLOCK) {
        const int64_t key = tmp[r];
        int rank = 0;
        for (int64_t j = 0; j < K; ++j) rank += tmp[j] < key ? 1 : 0;          // keys are distinct
        int64_t lo = 0, hi = ns;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (S.col_keys[mid] < key) lo = mid + 1; else hi = mid; }
        S.pKey[rank] = key; S.pIdx[rank] = (uint32_t)(ns + r); S.pLb[rank] = (uint32_t)lo;
    }
    __syncthreads();
    S.n_pend = (int)K;
}

// _pos_of_partition_end  src/pcsr.jl:177-186
__device__ int64_t d_partition_end(Seq& S, int64_t partition) {
    if (S.n_pend > 0) {            // no tombstones in this mode: the successor in key order, sorted part or pending list
        const int64_t kp = S.col_keys[partition - 1];
        int64_t sorted_succ = partition;                        // 0-based index of the next sorted entry
        if (partition > S.n_sorted) {
            const DFoundKey f = d_find_table_fast(S.col_keys, S.col_live, S.n_sorted, kp);
            sorted_succ = f.has ? f.pos : 0;                    // number of sorted keys < kp
        }
        const int64_t sidx = succ_index(S, sorted_succ, kp);
        return sidx >= 0 ? S.sems[sidx] - 1 : S.capacity;
    }
    const int64_t next = d_next_live_sem(S.sems, partition, S.table_len);
    return next != 0 ? S.sems[next - 1] - 1 : S.capacity;
}

// setindex!(pcsc, value, key, partition)  src/pcsr.jl:294-339
__device__ int d_pcsc_set(Seq& S, double val, int64_t key, int64_t partition) {
    if (partition < 1) { S.err = E_BOUNDS; return SEQ_ERROR; }
    while (partition > S.table_len) {                   // _add_partitions!  :312-319
        const int r = d_addpartition_append(S);
        if (r) return r;
    }
    const int64_t from = S.sems[partition - 1];
    if (from == 0) { S.err = E_DELETED; return SEQ_ERROR; }
    const int64_t te0 = DSA_TICK();
    const int64_t to = d_partition_end(S, partition);
    S.prof[11] += DSA_TICK() - te0;
    return d_set_in_range(S, key, val, from + 1, to, from);
}

// purge!(array, from, to)  src/writes.jl:80-91: clears every slot of [from, to]; returns the number of cells deleted
__device__ int64_t blk_purge(Seq& S, int64_t from, int64_t to) {
    if (to < from) return 0;
    const int64_t nb = blk_count(S, from, to + 1);
    __syncthreads();
    const int64_t lo0 = from - 1, hi0 = to - 1;
    const int64_t w0 = lo0 >> 6, w1 = hi0 >> 6;
    for (int64_t w = w0 + threadIdx.x; w <= w1; w += SEQ_BLOCK) S.occ[w] &= ~word_range_mask(w, lo0, hi0);
    __syncthreads();
    return nb;
}

// deletepartition!  src/pcsr.jl:188-204
__device__ int d_deletepartition(Seq& S, int64_t partition) {
    if (!(1 <= partition && partition <= S.table_len)) { S.err = E_BOUNDS; return SEQ_ERROR; }
    S.nb_partitions -= 1;
    const int64_t sem_pos = S.sems[partition - 1];
    if (sem_pos == 0) { S.err = E_ASSERT; return SEQ_ERROR; }
    const int64_t end = d_partition_end(S, partition);
    const int64_t nb = blk_purge(S, sem_pos, end);
    if (threadIdx.x == 0) S.sems[partition - 1] = 0;
    __syncthreads();
    const int64_t mid = sem_pos + (end - sem_pos) / 2;
    if (nb > 0) {
        S.nb_elements -= nb;
        return d_after_count_change(S, mid);
    }
    return 0;
}

__device__ int d_exec(Seq& S, const Op& op) {
    if (op.kind != OP_MPCSC_SET && S.n_pend > 0) d_merge_pending(S);
    switch (op.kind) {
        case OP_VEC_SET:
            return d_set_in_range(S, op.a, op.v, 1, S.capacity, 1);
        case OP_PCSC_SET:
            return d_pcsc_set(S, op.v, op.a, op.b);
        case OP_MPCSC_SET: {       // setindex!(mpcsc, value, row, col)  src/pcsr.jl:341-351
            if (S.n_pend == PEND_MAX) d_merge_pending(S);
            if (S.n_pend == 0) S.n_sorted = S.table_len;
            const int64_t tp0 = DSA_TICK();
            const bool no_tombstone = S.nb_partitions == S.table_len;
            const DFoundKey f = d_find_table_fast(S.col_keys, S.col_live, S.n_sorted, op.b, no_tombstone);
            int64_t col_pos = f.pos;
            bool found = f.has && f.key == op.b;
            int pj = 0;
            // successor of the column in key order (0-based table index, -1: none, -2: not known yet) — bounds its slot range
            int64_t sidx = -2;
            if (found && S.n_pend > 0) sidx = succ_index(S, col_pos, op.b);
            if (!found && S.n_pend > 0) {
                pj = pend_lower_bound(S, op.b);
                if (pj < S.n_pend && S.pKey[pj] == op.b) {
                    found = true; col_pos = (int64_t)S.pIdx[pj] + 1;
                    sidx = succ_index(S, (int64_t)S.pLb[pj], op.b);
                }
            }
            const int64_t tp1 = DSA_TICK();
            S.prof[0] += tp1 - tp0; S.prof[4] += 1;
            if (!found) {
                if (S.n_pend == 0 && (f.pos == S.table_len || !no_tombstone)) {
                    // the literal paths of the reference: append behind the last column, or middle insert with tombstones around
                    if (f.pos == S.table_len) {
                        if (S.table_len + 1 > S.table_cap) return SEQ_Y_TABLE_GROW | RERUN;
                        __syncthreads();
                        if (threadIdx.x == 0) { S.col_keys[S.table_len] = op.b; S.col_live[S.table_len] = 1; }
                        __syncthreads();
                        const int r = d_addpartition_append(S);
                        S.n_sorted = S.table_len;
                        if (r) return r;
                        col_pos = S.table_len;
                    } else {
                        const int r = d_addpartition_middle(S, f.pos, true, op.b);
                        S.n_sorted = S.table_len;
                        if (r) return r;
                        col_pos = f.pos + 1;
                    }
                } else {
                    // deferred middle insert: same semaphore cell at the same place, table entry at the end (see above)
                    if (S.table_len + 1 > S.table_cap) return SEQ_Y_TABLE_GROW | RERUN;
                    const int64_t lb = f.has ? f.pos : 0;
                    sidx = succ_index(S, lb, op.b);
                    const int64_t sem_pos = sidx >= 0 ? S.sems[sidx] - 1 : S.capacity;
                    const int64_t idx = S.table_len;
                    __syncthreads();
                    if (threadIdx.x == 0) { S.col_keys[idx] = op.b; S.col_live[idx] = 1; S.sems[idx] = sem_pos; }
                    __syncthreads();
                    S.table_len += 1;
                    S.nb_partitions += 1;
                    pend_insert(S, op.b, idx, lb, pj);
                    const int64_t ip = d_insert_after(S, SEM_KEY, (double)(idx + 1), sem_pos);
                    if (ip == 0) return SEQ_ERROR;
                    if (threadIdx.x == 0) S.sems[idx] = ip;
                    __syncthreads();
                    S.nb_elements += 1;
                    const int r = d_after_count_change(S, ip);
                    if (r) return r | RERUN;
                    col_pos = idx + 1;
                }
                S.prof[1] += DSA_TICK() - tp1; S.prof[5] += 1;
            }
            const int64_t tp2 = DSA_TICK();
            int rr;
            if (sidx == -2) rr = d_pcsc_set(S, op.v, op.a, col_pos);
            else {
                // setindex!(pcsc, value, key, partition) with the range end already known  src/pcsr.jl:294-310
                const int64_t from = S.sems[col_pos - 1];
                const int64_t to = sidx >= 0 ? S.sems[sidx] - 1 : S.capacity;
                rr = d_set_in_range(S, op.a, op.v, from + 1, to, from);
            }
            S.prof[2] += DSA_TICK() - tp2;
            return rr;
        }
        case OP_DELETE_PARTITION:
            return d_deletepartition(S, op.b);
        case OP_MPCSC_DELETECOLUMN: {   // deletecolumn!(mpcsc, col)  src/pcsr.jl:206-212
            const DFoundKey f = d_find_table_fast(S.col_keys, S.col_live, S.table_len, op.b);
            if (!(f.has && f.key == op.b)) { S.err = E_ARG; return SEQ_ERROR; }
            __syncthreads();
            if (threadIdx.x == 0) S.col_live[f.pos - 1] = 0;
            __syncthreads();
            return d_deletepartition(S, f.pos);
        }
        default:
            S.err = E_ARG;
            return SEQ_ERROR;
    }
}

__global__ __launch_bounds__(SEQ_BLOCK) void k_sequencer(KeyArr keys, double* vals, uint64_t* occ, int64_t* sems,
                                                         int64_t* col_keys, uint8_t* col_live, Ctl* ctl,
                                                         const Op* ops, int64_t n_ops, int64_t n_avail, int run_ok, const uint64_t* breaks,
                                                         Ctl* host_ctl, unsigned long long* host_seq, unsigned int seq) {
    // host_ctl / host_seq (pinned, may be null): the kernel hands its control block back itself — every word, then `seq` into the word the
    // host polls for — instead of a publish launch behind it (as parbatch.hip's k_publish does for a burst of rounds)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int64_t sRed[SEQ_BLOCK / 64];
    __shared__ uint32_t sWordOff[SMALL_W / 64 + 1];
    Seq S;
    S.keys = keys; S.vals = vals; S.occ = occ; S.sems = sems; S.col_keys = col_keys; S.col_live = col_live; S.ctl = ctl;
    S.capacity = ctl->capacity; S.seg = ctl->segment_capacity; S.height = ctl->height;
    S.nb_elements = ctl->nb_elements; S.nb_partitions = ctl->nb_partitions;
    S.table_len = ctl->table_len; S.table_cap = ctl->table_cap;
    S.stat_window_slots = ctl->stat_window_slots; S.stat_rebalances = ctl->stat_rebalances;
    S.stat_small = ctl->stat_small_rebalances;
    S.y_ws = S.y_we = S.y_m = 0; S.err = 0; S.tail_hint = false; S.breaks = breaks;
    __shared__ int64_t sPKey[PEND_MAX];
    __shared__ uint32_t sPIdx[PEND_MAX];
    __shared__ uint32_t sPLb[PEND_MAX];
    S.n_sorted = S.table_len; S.n_pend = 0; S.pKey = sPKey; S.pIdx = sPIdx; S.pLb = sPLb;
    for (int q = 0; q < 16; ++q) S.prof[q] = 0;
    const int64_t tk0 = DSA_TICK();
    S.sK = reinterpret_cast<int64_t*>(lds);
    S.sV = reinterpret_cast<double*>(lds + SMALL_W * sizeof(int64_t));
    S.sWordOff = sWordOff; S.sRed = sRed;
    __shared__ int64_t sLo[MAX_LEVELS], sHi[MAX_LEVELS];
    if (threadIdx.x < MAX_LEVELS) { sLo[threadIdx.x] = ctl->lo[threadIdx.x]; sHi[threadIdx.x] = ctl->hi[threadIdx.x]; }
    S.lo = sLo; S.hi = sHi;
    __syncthreads();
    if (col_keys != nullptr && ctl->n_pending > 0) d_import_pending(S, ctl->n_pending);

    int64_t i = ctl->next_op;
    int status = SEQ_DONE;
    Op op = ops[i < n_ops ? i : 0];
    int64_t run_cooldown = 0;
    const int64_t no_run_at = ctl->no_run_at;
    for (; i < n_ops; ++i) {
        const Op nxt = ops[i + 1 < n_ops ? i + 1 : i];          // prefetched under the current op's memory traffic
        const bool vec_cand = sems == nullptr && op.kind == OP_VEC_SET && nxt.kind == OP_VEC_SET && nxt.a > op.a;
        const bool csc_cand = col_keys != nullptr && op.kind == OP_MPCSC_SET && nxt.kind == OP_MPCSC_SET &&
                              (nxt.b > op.b || (nxt.b == op.b && nxt.a > op.a));
        // (no detection while middle inserts are pending: an append run needs its columns behind the last key)
        if (run_ok && (vec_cand || csc_cand) && op.v != 0.0 && i + RUN_MIN <= n_avail && i != no_run_at && S.n_pend == 0 && --run_cooldown < 0) {
            const int64_t R = vec_cand ? d_detect_append_run(S, ops, i, n_avail) : d_detect_pcsc_run(S, ops, i, n_avail);
            if (R > 0) {
                S.y_ws = i; S.y_we = S.nb_elements; S.y_m = R;
                status = SEQ_Y_APPEND_RUN;
                break;
            }
            run_cooldown = 32;
        }
        const int r = d_exec(S, op);
        if (r != 0) {
            status = r & 0xff;
            if (!(r & RERUN) && status != SEQ_ERROR) ++i;      // the op itself is complete once the host has acted
            break;
        }
        op = nxt;
    }
    // pending entries stay at the end of the tables (Ctl::n_pending): the host merges them with the whole chip (tables.hip) when
    // enough have piled up and before the batch returns; the next launch re-imports what is left
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl->next_op = i;
        ctl->status = status;
        ctl->err = S.err;
        ctl->err_op = (status == SEQ_ERROR) ? i : -1;
        ctl->nb_elements = S.nb_elements; ctl->nb_partitions = S.nb_partitions; ctl->table_len = S.table_len;
        ctl->n_pending = S.n_pend;
        ctl->y_ws = S.y_ws; ctl->y_we = S.y_we; ctl->y_m = S.y_m;
        ctl->stat_window_slots = S.stat_window_slots; ctl->stat_rebalances = S.stat_rebalances;
        ctl->stat_small_rebalances = S.stat_small;
        S.prof[7] = DSA_TICK() - tk0;
        for (int q = 0; q < 16; ++q) ctl->prof[q] += S.prof[q];
    }
    if (host_seq != nullptr) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();                               // thread 0's words are on their way to the L2
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(ctl);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(host_ctl);
        for (int q = threadIdx.x; q < (int)(sizeof(Ctl) / 8); q += SEQ_BLOCK)
            __hip_atomic_store(dst + q, __hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(host_seq, (unsigned long long)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_sequencer(KeyArr keys, double* vals, uint64_t* occ, int64_t* sems, int64_t* col_keys,
                            uint8_t* col_live, Ctl* ctl, const Op* ops, int64_t n_ops, int64_t n_avail, bool run_ok, const uint64_t* breaks,
                            Ctl* host_ctl, unsigned long long* host_seq, unsigned int seq, hipStream_t stream) {
    const size_t lds_bytes = (size_t)SMALL_W * (sizeof(int64_t) + sizeof(double));
    static PerDeviceOnce once;
    {
        hipError_t e = once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(k_sequencer), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_sequencer, dim3(1), dim3(SEQ_BLOCK), lds_bytes, stream, keys, vals, occ, sems, col_keys, col_live,
                       ctl, ops, n_ops, n_avail, run_ok ? 1 : 0, breaks, host_ctl, host_seq, seq);
    return hipGetLastError();
}

// ---- batched read-only lookups -----------------------------------------------------------------------
// getindex(pma,key) src/pma.jl:189-193 ; getindex(pcsc,key,partition) src/pcsr.jl:222-232 ;
// getindex(mpcsc,row,col) src/pcsr.jl:261-267.  One lane per query; each lane replays the reference's bisection.
// one lookup by one lane; *err receives E_BOUNDS / E_ASSERT (0: none)
__device__ double get_one(int mode, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, const int64_t* sems,
                          const int64_t* col_keys, const uint8_t* col_live, int64_t table_len, int64_t key, int64_t qb, int32_t* err) {
    int64_t from = 1, to = capacity;
    if (mode != 0) {
        int64_t partition = qb;
        if (mode == 2) {
            const DFoundKey f = d_find_table(col_keys, col_live, table_len, qb);
            if (!(f.has && f.key == qb)) return 0.0;
            partition = f.pos;
        }
        if (partition < 1 || partition > table_len) { *err = E_BOUNDS; return 0.0; }
        from = sems[partition - 1];
        if (from == 0) { *err = E_ASSERT; return 0.0; }   // _pos_of_partition_start @assert
        const int64_t next = d_next_live_sem(sems, partition, table_len);
        to = next != 0 ? sems[next - 1] - 1 : capacity;
    }
    const DFound f = d_find(keys, vals, occ, key, from, to);
    return (f.has && f.key == key) ? f.val : 0.0;
}
__global__ void k_get_batch(int mode, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                            const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live, int64_t table_len,
                            const int64_t* qa, const int64_t* qb, int64_t n, double* out, int32_t* err_out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t err = 0;
    out[i] = get_one(mode, keys, vals, occ, capacity, sems, col_keys, col_live, table_len, qa[i], mode != 0 ? qb[i] : 0, &err);
    if (err) atomicCAS(err_out, 0, err);
}
// Up to 64 lookups without a copy command: the queries are read from, and the answers written to, a pinned landing area of the handle —
// io[0..63] keys, io[64..127] partitions / columns, io[128..191] answers, io[192] first error, io[193] the sequence number the host polls
// for.  A scalar getindex (A[i, j], v[k]) is one launch and a poll: ~15 us instead of ~70 (two uploads, a memset, two downloads into pageable
// memory, a stream synchronisation).
__global__ __launch_bounds__(64) void k_get_small(int mode, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                                                  const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live, int64_t table_len,
                                                  int64_t* io, int n, unsigned long long seq) {
    const int lane = threadIdx.x;
    int32_t err = 0;
    if (lane < n) {
        const int64_t key = __hip_atomic_load(io + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const int64_t qb = mode != 0 ? __hip_atomic_load(io + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
        const double v = get_one(mode, keys, vals, occ, capacity, sems, col_keys, col_live, table_len, key, qb, &err);
        __hip_atomic_store(io + 128 + lane, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint64_t eb = __ballot(err != 0);
    const int first = eb ? __ffsll((unsigned long long)eb) - 1 : 0;
    const int32_t e0 = __shfl(err, first, 64);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        __hip_atomic_store(io + 192, (int64_t)(eb ? e0 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(io) + 193, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
hipError_t launch_get_small(int mode, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, const int64_t* sems,
                            const int64_t* col_keys, const uint8_t* col_live, int64_t table_len, int64_t* io, int n, unsigned long long seq,
                            hipStream_t stream) {
    hipLaunchKernelGGL(k_get_small, dim3(1), dim3(64), 0, stream, mode, keys, vals, occ, capacity, sems, col_keys, col_live, table_len, io, n, seq);
    return hipGetLastError();
}

hipError_t launch_get_batch(int mode, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                            const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live, int64_t table_len,
                            const int64_t* qa, const int64_t* qb, int64_t n, double* out, int32_t* err_out,
                            hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int block = 64;
    hipLaunchKernelGGL(k_get_batch, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, stream, mode, keys, vals, occ,
                       capacity, sems, col_keys, col_live, table_len, qa, qb, n, out, err_out);
    return hipGetLastError();
}

// ---- device-side invariant checker (the structural checks of the reference's test/utils.jl:68-113, at full size) ------
// report[0] = occupied cells, [1] = semaphore cells, [2] = semaphore cells whose table entry does not point back,
// [3] = key-order violations inside a partition (or anywhere, for a plain vector), [4] = live table entries that do not
// point at a semaphore cell holding their id, [5] = occupancy bits at or beyond the capacity
__global__ void k_check_slots(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, int64_t occ_words,
                              const int64_t* sems, int64_t table_len, unsigned long long* report) {
    const int64_t s0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;      // 0-based slot
    unsigned long long n_occ = 0, n_sem = 0, bad_sem = 0, bad_order = 0;
    if (s0 < capacity && ((occ[s0 >> 6] >> (s0 & 63)) & 1ull)) {
        n_occ = 1;
        const int64_t k = keys[s0];
        if (sems != nullptr && k == SEM_KEY) {
            n_sem = 1;
            const int64_t id = (int64_t)vals[s0];
            if (id < 1 || id > table_len || sems[id - 1] != s0 + 1) bad_sem = 1;
        } else {
            const int64_t p = d_prev_occupied(occ, s0, 1);                  // previous occupied position (1-based), 0 if none
            if (p >= 1) {
                const int64_t pk = keys[p - 1];
                const bool prev_is_sem = sems != nullptr && pk == SEM_KEY;
                if (!prev_is_sem && !(pk < k)) bad_order = 1;
            }
        }
    }
    unsigned long long beyond = 0;
    const int64_t w = s0;                                                   // reuse the grid for the words past the capacity
    if (w < occ_words && (w << 6) >= capacity) beyond = (unsigned long long)popc64(occ[w]);
    else if (w < occ_words && ((w + 1) << 6) > capacity) beyond = (unsigned long long)popc64(occ[w] & ~mask_lt((int)(capacity & 63)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n_occ += __shfl_xor(n_occ, o, 64); n_sem += __shfl_xor(n_sem, o, 64); bad_sem += __shfl_xor(bad_sem, o, 64);
        bad_order += __shfl_xor(bad_order, o, 64); beyond += __shfl_xor(beyond, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_occ) atomicAdd(&report[0], n_occ);
        if (n_sem) atomicAdd(&report[1], n_sem);
        if (bad_sem) atomicAdd(&report[2], bad_sem);
        if (bad_order) atomicAdd(&report[3], bad_order);
        if (beyond) atomicAdd(&report[5], beyond);
    }
}
__global__ void k_check_table(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, const int64_t* sems,
                              const int64_t* col_keys, const uint8_t* col_live, int64_t table_len, unsigned long long* report) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= table_len) return;
    const int64_t pos = sems[i];
    bool bad = false;
    if (pos != 0) {
        if (pos < 1 || pos > capacity || !occ_test(occ, pos) || keys[pos - 1] != SEM_KEY || vals[pos - 1] != (double)(i + 1)) bad = true;
        if (col_live != nullptr && !col_live[i]) bad = true;
        if (col_live != nullptr && i > 0) {                                  // live column keys ascend with the id
            int64_t j = i - 1;
            while (j >= 0 && !col_live[j]) --j;
            if (j >= 0 && !(col_keys[j] < col_keys[i])) bad = true;
        }
    } else if (col_live != nullptr && col_live[i]) bad = true;
    if (bad) atomicAdd(&report[4], 1ull);
}
hipError_t launch_check(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity, int64_t occ_words,
                        const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live, int64_t table_len,
                        unsigned long long* report, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(report, 0, 8 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const int64_t n = capacity > occ_words ? capacity : occ_words;
    hipLaunchKernelGGL(k_check_slots, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, keys, vals, occ, capacity, occ_words,
                       sems, table_len, report);
    if (sems != nullptr && table_len > 0)
        hipLaunchKernelGGL(k_check_table, dim3((unsigned)((table_len + 255) / 256)), dim3(256), 0, stream, keys, vals, occ, capacity,
                           sems, col_keys, col_live, table_len, report);
    return hipGetLastError();
}

// view(mpcsc, :, col)  src/views.jl:15-35 : slot range of the column
__global__ void k_partition_range(const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live, int64_t table_len,
                                  int64_t capacity, int64_t col, int64_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = 0; out[1] = 0; out[2] = 0;
    const DFoundKey f = d_find_table(col_keys, col_live, table_len, col);
    if (!(f.has && f.key == col)) return;
    const int64_t from = sems[f.pos - 1];
    if (from == 0) { out[2] = E_ASSERT; return; }
    const int64_t next = d_next_live_sem(sems, f.pos, table_len);
    out[0] = from + 1;
    out[1] = next != 0 ? sems[next - 1] - 1 : capacity;
    out[2] = 0;
    out[3] = f.pos;
}
// view(mpcsc, :, col) (src/views.jl:15-35) in ONE launch for partitions of up to VIEW_SMALL_SLOTS slots: partition lookup, then the
// occupied cells of its slot range packed in slot order into out_k / out_v.  meta[0] = from, meta[1] = to, meta[2] = error code,
// meta[3] = partition id, meta[4] = number of cells or -1 when the range is longer (the caller then takes the general K-pack path),
// meta[5] = key of the last cell packed (host word [6]).  Column views, slices and deletecolumn! / deleterow! need one host round trip
// this way.  One workgroup of VIEW_BLOCK threads (round 6; one wave until then: a 10 000-cell column took 115 us of dependent word after
// word): (1) every thread counts its share of the occupancy words, all loads in flight at once; (2) exclusive prefix per word in LDS;
// (3) the waves take the words round-robin, lane <-> slot, and store rank-addressed — coalesced reads, independent of each other.
constexpr int VIEW_BLOCK = 1024;
constexpr int VIEW_WORDS = (int)(VIEW_SMALL_SLOTS / 64) + 1;          // a range of VIEW_SMALL_SLOTS slots touches at most this many words
__global__ __launch_bounds__(VIEW_BLOCK) void k_view_small(KeyArr keys, const double* __restrict__ vals,
                                                   const uint64_t* __restrict__ occ, const int64_t* sems, const int64_t* col_keys,
                                                   const uint8_t* col_live, int64_t table_len, int64_t capacity, int64_t col,
                                                   KeyArr out_k, double* __restrict__ out_v, int64_t out_cap, int64_t* meta,
                                                   int64_t* host, int64_t host_cells, unsigned long long seq,
                                                   int64_t range_from, int64_t range_to) {
    // host (pinned, may be null): [0..4] the meta words, [5] the sequence number the host polls for, [6] the last key, then host_cells
    // keys and host_cells values — the first cells of the view go straight to the host with the meta words: one launch and no copy
    // command for a short column (a D2H copy into the caller's pageable vectors + a stream synchronisation cost 60 us per view; 20 us this way)
    __shared__ uint64_t sMask[VIEW_WORDS];
    __shared__ uint32_t sPref[VIEW_WORDS];
    __shared__ uint32_t sWave[VIEW_BLOCK / 64];
    __shared__ int64_t sLast;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t from = 0, to = 0, err = 0, pid = 0;
    // range_from > 0: no partition lookup, the stored cells of the slot range [range_from, range_to] (iteration over a small vector,
    // src/pma.jl:165-180: the same one-launch hand-over as a column view).  The lookup is executed by every wave (uniform result).
    DFoundKey f{0, 0, false};
    if (range_from > 0) { from = range_from; to = range_to; }
    else f = d_find_table_fast(col_keys, col_live, table_len, col);
    if (range_from <= 0 && f.has && f.key == col) {
        const int64_t sp = sems[f.pos - 1];
        if (sp == 0) err = E_ASSERT;
        else {
            const int64_t next = d_next_live_sem(sems, f.pos, table_len);
            from = sp + 1;
            to = next != 0 ? sems[next - 1] - 1 : capacity;
            pid = f.pos;
        }
    }
    int64_t cnt = 0;
    if (threadIdx.x == 0) sLast = 0;
    if (from != 0 && to >= from) {
        if (to - from + 1 > VIEW_SMALL_SLOTS) cnt = -1;
        else {
            const int64_t lo0 = from - 1, hi0 = to - 1;
            const int64_t w0 = lo0 >> 6;
            const int nw = (int)((hi0 >> 6) - w0) + 1;                      // <= VIEW_WORDS
            const int per = (nw + VIEW_BLOCK - 1) / VIEW_BLOCK;             // words per thread, contiguous
            const int t0 = threadIdx.x * per;
            uint32_t mine = 0;
            for (int q = t0; q < t0 + per && q < nw; ++q) {
                const uint64_t m = occ[w0 + q] & word_range_mask(w0 + q, lo0, hi0);
                sMask[q] = m;
                mine += (uint32_t)popc64(m);
            }
            // exclusive scan of the per-thread counts: inside the wave by shuffles, across the waves through LDS
            uint32_t inc = mine;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
            if (lane == 63) sWave[wave] = inc;
            __syncthreads();
            uint32_t base = 0, total = 0;
            for (int v = 0; v < VIEW_BLOCK / 64; ++v) { const uint32_t c = sWave[v]; if (v < wave) base += c; total += c; }
            uint32_t run = base + inc - mine;
            for (int q = t0; q < t0 + per && q < nw; ++q) { sPref[q] = run; run += (uint32_t)popc64(sMask[q]); }
            __syncthreads();
            cnt = (int64_t)total;
            if (cnt > out_cap) cnt = -1;
            else {
                // four words per wave and step: the loads of all four are in flight before the first store (the stores may alias the
                // loads as far as the compiler knows: word after word, every iteration waited for its own loads — 35 us for 366 words)
                constexpr int NWAVES = VIEW_BLOCK / 64, U = 4;
                for (int q0 = wave; q0 < nw; q0 += NWAVES * U) {
                    int64_t kk[U]; double vv[U]; int64_t r[U]; bool on[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int q = q0 + u * NWAVES;
                        on[u] = false; r[u] = 0; kk[u] = 0; vv[u] = 0.0;
                        if (q < nw) {
                            const uint64_t mask = sMask[q];
                            if ((mask >> lane) & 1ull) {
                                on[u] = true;
                                r[u] = (int64_t)sPref[q] + popc64(mask & mask_lt(lane));
                                const int64_t slot = ((w0 + q) << 6) + lane;
                                kk[u] = keys[slot]; vv[u] = vals[slot];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (!on[u]) continue;
                        out_k[r[u]] = kk[u]; out_v[r[u]] = vv[u];
                        if (r[u] == cnt - 1) sLast = kk[u];
                        if (host != nullptr && r[u] < host_cells) {
                            __hip_atomic_store(host + 8 + r[u], kk[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(host + 8 + host_cells + r[u], __double_as_longlong(vv[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);                     // every lane's cells have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t last_key = cnt > 0 ? sLast : 0;
        meta[0] = from; meta[1] = to; meta[2] = err; meta[3] = pid; meta[4] = cnt; meta[5] = last_key;
        if (host != nullptr) {
            const int64_t m5[5] = {from, to, err, pid, cnt};
            for (int q = 0; q < 5; ++q) __hip_atomic_store(host + q, m5[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host + 6, last_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(host) + 5, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
hipError_t launch_view_small(KeyArr keys, const double* vals, const uint64_t* occ, const int64_t* sems, const int64_t* col_keys,
                             const uint8_t* col_live, int64_t table_len, int64_t capacity, int64_t col, KeyArr out_k, double* out_v,
                             int64_t out_cap, int64_t* meta, int64_t* host, int64_t host_cells, unsigned long long seq, int64_t range_from,
                             int64_t range_to, hipStream_t stream) {
    hipLaunchKernelGGL(k_view_small, dim3(1), dim3(VIEW_BLOCK), 0, stream, keys, vals, occ, sems, col_keys, col_live, table_len, capacity, col, out_k,
                       out_v, out_cap, meta, host, host_cells, seq, range_from, range_to);
    return hipGetLastError();
}

hipError_t launch_partition_range(const int64_t* sems, const int64_t* col_keys, const uint8_t* col_live,
                                  int64_t table_len, int64_t capacity, int64_t col, int64_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_partition_range, dim3(1), dim3(64), 0, stream, sems, col_keys, col_live, table_len, capacity, col, out);
    return hipGetLastError();
}

// ---- parity hooks (include/dsa.h: dsa_dbg_raw_*) ------------------------------------------------------------------------------
// The slot-array primitives of the sequencer run on a caller-supplied RAW slot array (any length, any content), so that the
// reference's own unit-test vectors (test/unit/finds.jl, test/unit/writes.jl) are checked on the device code itself and not only on
// the CPU oracle.  One workgroup, one op.  out[0] = error code, out[1] = position, out[2] = flag (element found / new key / deleted),
// out[3] = key of the element found, out[4] = its value (bits), out[5] = cells purged.
__global__ __launch_bounds__(SEQ_BLOCK) void k_dbg_raw(KeyArr keys, double* vals, uint64_t* occ, int64_t* sems, int64_t len, int op,
                                                       int64_t key, double val, int64_t from, int64_t to, int64_t m, int64_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int64_t sRed[SEQ_BLOCK / 64];
    __shared__ uint32_t sWordOff[SMALL_W / 64 + 1];
    Seq S;
    S.keys = keys; S.vals = vals; S.occ = occ; S.sems = sems; S.col_keys = nullptr; S.col_live = nullptr; S.ctl = nullptr;
    S.capacity = len; S.seg = 1; S.height = 0; S.nb_elements = 0; S.nb_partitions = 0; S.table_len = 0; S.table_cap = 0;
    S.stat_window_slots = S.stat_rebalances = S.stat_small = 0;
    S.y_ws = S.y_we = S.y_m = 0; S.err = 0; S.tail_hint = false;
    S.n_sorted = 0; S.n_pend = 0; S.pKey = nullptr; S.pIdx = nullptr; S.pLb = nullptr;
    for (int q = 0; q < 16; ++q) S.prof[q] = 0;
    S.sK = reinterpret_cast<int64_t*>(lds);
    S.sV = reinterpret_cast<double*>(lds + SMALL_W * sizeof(int64_t));
    S.sWordOff = sWordOff; S.sRed = sRed; S.lo = nullptr; S.hi = nullptr;
    int64_t r_pos = 0, r_flag = 0, r_key = 0, r_nb = 0; double r_val = 0.0;
    switch (op) {
        case DBG_FIND: case DBG_FIND_FAST: {                       // find(array, key, from, to)  src/finds.jl:29-57
            const DFound f = op == DBG_FIND ? d_find(keys, vals, occ, key, from, to) : d_find_fast(keys, vals, occ, key, from, to);
            r_pos = f.pos; r_flag = f.has ? 1 : 0; r_key = f.key; r_val = f.val;
            break;
        }
        case DBG_INSERT: case DBG_INSERT_FAST: {                   // insert!(array, key, value, from, to, semaphores)  src/writes.jl:14-43
            const DFound f = op == DBG_INSERT ? d_find(keys, vals, occ, key, from, to) : d_find_fast(keys, vals, occ, key, from, to);
            if (f.has && f.key == key && from <= f.pos && f.pos <= to) {
                __syncthreads();
                if (threadIdx.x == 0) vals[f.pos - 1] = val;
                r_pos = f.pos; r_flag = 0;
            } else {
                r_pos = d_insert_after(S, key, val, f.pos);
                r_flag = 1;
            }
            break;
        }
        case DBG_DELETE: case DBG_DELETE_FAST: {                   // delete!(array, key, from, to)  src/writes.jl:57-68
            const DFound f = op == DBG_DELETE ? d_find(keys, vals, occ, key, from, to) : d_find_fast(keys, vals, occ, key, from, to);
            if (f.has && f.key == key) {
                __syncthreads();
                if (threadIdx.x == 0) occ_clear(S, f.pos);
                r_pos = f.pos; r_flag = 1;
            }
            break;
        }
        case DBG_PURGE:                                            // purge!(array, from, to)  src/writes.jl:80-91
            if (to >= from) { r_nb = blk_purge(S, from, to); r_pos = from + (to - from) / 2; }
            break;
        case DBG_REBALANCE:                                        // pack! + spread! of [from, to] holding m cells  src/moves.jl:94-171
            blk_rebalance_small(S, from, to, m);
            break;
        default:
            S.err = E_ARG;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = S.err; out[1] = r_pos; out[2] = r_flag; out[3] = r_key; out[4] = __double_as_longlong(r_val); out[5] = r_nb;
    }
}

hipError_t launch_dbg_raw_block(KeyArr keys, double* vals, uint64_t* occ, int64_t* sems, int64_t len, int op, int64_t key, double val,
                                int64_t from, int64_t to, int64_t m, int64_t* out, hipStream_t stream) {
    const size_t lds_bytes = (size_t)SMALL_W * (sizeof(int64_t) + sizeof(double));
    static PerDeviceOnce once;
    {
        hipError_t e = once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(k_dbg_raw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_dbg_raw, dim3(1), dim3(SEQ_BLOCK), lds_bytes, stream, keys, vals, occ, sems, len, op, key, val, from, to, m, out);
    return hipGetLastError();
}

}  // namespace dsa
