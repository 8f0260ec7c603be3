// csrc/appendmodel.hip — count-only replay of an append run ("model v3") for gfx950.
//
// A run of setindex! calls with ascending keys above the last key (src/pma.jl:196-213, src/pcsr.jl:294-351 with
// addpartition!(pcsc) src/pcsr.jl:99-112) inserts every cell into the LAST leaf of the array (_insert! behind the tail or, for a
// tail on the last slot / a semaphore, with a shift to the left inside the leaf, src/writes.jl:26-43) as long as that leaf keeps a
// free slot besides the last one — which the leaf threshold guarantees for segments of 16 slots and more.  The windows
// _look_for_rebalance! (src/pma.jl:105-141) consults are then the SUFFIXES of the array, one per level, and its decisions depend on
// nothing but their cell counts: an append adds one to every suffix count; the first level k >= 1 whose count c lies inside its
// integer thresholds is rebalanced ("event (k, c)"), which resets the counts below it to the closed form of spread!
// (src/moves.jl:120-171: cells in the last W_i offsets of c cells spread over W_k slots); no level -> _extend!, the run ends in front
// of that op.  Positions of the cells inside the last leaf matter only for the ops behind the last event (at most seg - 2 of them):
// those are replayed bit by bit at the end.
//
// The process is hierarchical: between two visits of level k ("every level below k rejects") nothing outside the level-k suffix is
// read or written, and what happens inside is a function of (k, c) alone.  So, per level k <= Lp and count c:
//     V_k[c] = appends (and rebalances, window slots) from event (k, c) to the next visit of level k
//     X_k[c] = the same summed along the chain c -> c + V_k[c] -> ... while level k accepts   (0 when it rejects c)
// are TABLES in LDS, filled level by level by the whole workgroup (one thread per entry); one wave then drives the levels above Lp
// event by event — a descent through the tables gives the time of the next visit of level Lp + 1, lanes <-> levels evaluate the
// thresholds and the closed-form counts — and a final descent finds the last rebalance of every level that no wider one followed.
// The bitmap is written once: those patterns, widest first, then the last leaf.  Nothing is committed before that point: any
// precondition that fails (counts outside the tables, a leaf without its free slots) leaves the bitmap untouched and the run to
// k_append_run (sequencer.hip), which also takes over whatever this kernel did not consume.
#include "dsa_dev.h"

namespace dsa {

namespace {

constexpr int M3_THREADS = 1024;
constexpr int M3_X_ENTRIES = 10240, M3_V_ENTRIES = 5120;        // dynamic LDS: X 80 KB + one level of V 40 KB + the step of every V entry 20 KB
constexpr int M3_MAX_TABLE_W = 65536;                           // (16-bit times in the entries: checked when they are packed)
constexpr int64_t M3_MIN_RUN = 512;                             // shorter runs: the per-op replay is cheaper than filling the tables

// entry: [15:0] appends  [31:16] rebalances  [63:32] window slots
__device__ __forceinline__ uint64_t m3_pack(uint32_t dt, uint32_t reb, uint64_t slots) { return (uint64_t)dt | ((uint64_t)reb << 16) | (slots << 32); }
__device__ __forceinline__ int m3_dt(uint64_t e) { return (int)(e & 0xffffu); }
__device__ __forceinline__ uint32_t m3_reb(uint64_t e) { return (uint32_t)((e >> 16) & 0xffffu); }
__device__ __forceinline__ uint64_t m3_slots(uint64_t e) { return e >> 32; }

struct M3Shared {
    int32_t W[64], lo[64], hi[64], cmin[64], base[64], n[64];
    int32_t cnt[64];                       // suffix cell count of level k (the driver's state)
    int32_t ev_c[64];                      // last rebalance of level k that no wider one followed: its cell count
    int32_t tauL[64]; uint32_t rebL[64]; unsigned long long slL[64];
    unsigned long long bk[40];             // popcount buckets of the suffix-count pass
    unsigned long long ev_mask;
    unsigned long long reb, slots;
    int32_t H, Lp, seg, bail, consumed, leaf_ops, ended, top_events;
    unsigned long long last_word;
};

__device__ __forceinline__ int ufl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t ufl64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// LDS traffic between the lanes of ONE wave: its LDS instructions execute in order, the fence keeps the compiler from moving them
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// cells in the last w offsets of spread!(c cells over W slots), g = make_geom(W, c)   (src/moves.jl:120-171)
__device__ __forceinline__ int m3_suffix_cells(const SpreadGeom& g, int W, int w) { return w - ((int)g.E - gaps_le(g, W - w)); }
// appends until the leaf rejects (every append visits the leaf): the count leaves [lo, hi]
__device__ __forceinline__ int m3_leaf_reject(int c0, int lo0, int hi0) { return (c0 + 1 < lo0 || c0 + 1 > hi0) ? 1 : hi0 - c0 + 1; }

// ---- the driver's state lives in the registers of wave 0, lane <-> level: constants of the level (window, thresholds, table range),
// its suffix count, the count of its surviving rebalance.  A level's value is broadcast with v_readlane (no memory round trip); the
// only LDS traffic of a step is the table lookup itself, whose address depends on the time accumulated so far.
struct M3Lane { int W, lo, hi, cmin, base; };
__device__ __forceinline__ int rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
struct M3Rec { int tau; uint32_t reb; uint64_t slots; };      // lane i: time of the first visit of level i, what the complete chains below added

// time of the first visit of level `upto` from the suffix counts of the levels below (tables of levels 1 .. upto-1); adds the
// rebalances / window slots of the complete chains below.  REC: per level into `rec`.  Wave-uniform.
template <bool REC>
__device__ __forceinline__ int m3_descent(const M3Lane& L, int cnt, const uint64_t* X, int upto, uint32_t& reb, uint64_t& slots, int& bail,
                                          int lane, M3Rec& rec) {
    int tau = m3_leaf_reject(rl(cnt, 0), rl(L.lo, 0), rl(L.hi, 0));
    if (REC && lane == 1) { rec.tau = tau; rec.reb = 0; rec.slots = 0; }
#pragma clang loop unroll(disable)
    for (int i = 1; i < upto; ++i) {
        const int ci = rl(cnt, i) + tau;
        if (ci >= rl(L.lo, i) && ci <= rl(L.hi, i)) {
            const int cm = rl(L.cmin, i);
            if (ci < cm) { bail = 10; return tau; }
            const uint64_t x = ufl64(X[rl(L.base, i) + ci - cm]);
            tau += m3_dt(x); reb += m3_reb(x); slots += m3_slots(x);
        }
        if (REC && lane == i + 1) { rec.tau = tau; rec.reb = reb; rec.slots = slots; }
    }
    return tau;
}

// event (k, c): the counts of the levels below k in closed form (lane <-> level); false: the last leaf lost its preconditions
__device__ __forceinline__ bool m3_reset_below(const M3Lane& L, int& cnt, int k, int c, int lane, int maxc0) {
    const int Wk = rl(L.W, k);
    SpreadGeom g;                                           // make_geom(Wk, c); E / W — only gaps_le's starting guess — by a multiply
    g.W = Wk; g.E = Wk - c;                                 // (W is a power of two: exact either way)
    g.f = g.E > 0 ? (double)Wk / (double)(Wk - c) : 0.0;
    g.inv_f = (double)(Wk - c) * (1.0 / (double)Wk);
    if (lane < k) cnt = m3_suffix_cells(g, Wk, L.W);
    const int c0 = rl(cnt, 0);
    return c0 >= 1 && c0 <= maxc0;
}

// the last b appends of the run from the suffix counts of the levels below kmax; level kmax is not visited by them.  Records the
// surviving rebalance of every level below kmax (evmask / evc) and returns the number of trailing ops that no level >= 1 follows.
// Per level m that is visited inside the budget: the chain of its events c -> c + V_m[c] is walked with the steps kept from the
// table fill (Vdt) up to the last event inside the budget; what the complete epochs before it add up to is X_m[first] - X_m[last].
__device__ int m3_final_descent(const M3Lane& L, int& cnt, const uint64_t* X, const uint16_t* Vdt, int kmax, int b, int lane, int maxc0,
                                uint64_t& reb_tot, uint64_t& slots_tot, uint64_t& evmask, int& evc, int& bail) {
    for (;;) {
        uint32_t r = 0; uint64_t s = 0;
        M3Rec rec; rec.tau = 0; rec.reb = 0; rec.slots = 0;
        m3_descent<true>(L, cnt, X, kmax, r, s, bail, lane, rec);
        if (bail) return 0;
        if (rl(rec.tau, 1) > b) return b;
        int m = 1;
        while (m + 1 <= kmax && rl(rec.tau, m + 1) <= b) ++m;
        if (m >= kmax) { bail = 11; return 0; }
        reb_tot += (uint64_t)(uint32_t)rl((int)rec.reb, m); slots_tot += rl64(rec.slots, m);
        const int tt = rl(rec.tau, m);
        const int c0 = rl(cnt, m) + tt;
        const int lom = rl(L.lo, m), him = rl(L.hi, m), Wm = rl(L.W, m), cm = rl(L.cmin, m), bm = rl(L.base, m);
        if (c0 < lom || c0 > him || c0 < cm) { bail = 12; return 0; }
        int c = c0;
        for (;;) {
            const int dv = ufl((int)Vdt[bm + c - cm]);
            if (tt + (c - c0) + dv > b) break;
            c += dv;
            if (c > him) { bail = 14; return 0; }          // (the chain ends behind the budget: cannot happen)
        }
        const uint64_t x0 = ufl64(X[bm + c0 - cm]), xf = ufl64(X[bm + c - cm]);
        reb_tot += (uint64_t)(m3_reb(x0) - m3_reb(xf)) + 1ull;
        slots_tot += (m3_slots(x0) - m3_slots(xf)) + (uint64_t)Wm;
        evmask = (evmask & ~((1ull << m) - 1ull)) | (1ull << m);
        if (lane == m) evc = c;
        if (!m3_reset_below(L, cnt, m, c, lane, maxc0)) { bail = 13; return 0; }
        b -= tt + (c - c0);
        kmax = m;
    }
}

// ---- suffix cell counts of every level into sh.cnt (whole workgroup; sh.bk zeroed by the caller): word r from the end falls into
//      bucket bits(r); the suffix of 2^j words is buckets 0..j.  Words r >= 1024 are read in rows of 1024 (one word per thread: a row
//      lies in ONE bucket), summed per thread while the bucket stays the same; the first 1024 words go to their buckets one by one
__device__ void m3_suffix_counts(M3Shared& sh, const uint64_t* occ, int64_t cap, int64_t seg, int H, int tid, int lane) {
    const int64_t nwords = cap >> 6;
    {
        if (tid < nwords) {
            const int b = tid == 0 ? 0 : 32 - __clz(tid);
            const unsigned long long c = (unsigned long long)popc64(occ[nwords - 1 - tid]);
            if (c) atomicAdd(&sh.bk[b], c);
        }
        unsigned long long acc = 0;
        int cur_b = 11;
        for (int64_t row = 1; row * (int)blockDim.x < nwords; ++row) {
            const int b = 64 - __clzll((long long)(row * (int)blockDim.x));           // bits(r) of every r in the row
            if (b != cur_b) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
                if (lane == 0 && acc) atomicAdd(&sh.bk[cur_b], acc);
                acc = 0; cur_b = b;
            }
            acc += (unsigned long long)popc64(occ[nwords - 1 - (row * (int)blockDim.x + tid)]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0 && acc) atomicAdd(&sh.bk[cur_b], acc);
    }
    __syncthreads();
    if (tid <= H) {
        const int64_t Wk = seg << tid;
        unsigned long long c = 0;
        if (Wk < 64) c = (unsigned long long)popc64(occ[nwords - 1] >> (64 - Wk));
        else {
            int j = 0;
            while ((64ll << j) < Wk) ++j;
            for (int b = 0; b <= j; ++b) c += sh.bk[b];
        }
        sh.cnt[tid] = (int32_t)c;
    }
    __syncthreads();
}

__global__ __launch_bounds__(M3_THREADS) void k_append_model3(uint64_t* occ, Ctl* ctl, int64_t R, const uint64_t* flags, const int64_t* d_T,
                                                              int64_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char m3_lds[];
    uint64_t* X = reinterpret_cast<uint64_t*>(m3_lds);
    uint64_t* V = X + M3_X_ENTRIES;
    uint16_t* Vdt = reinterpret_cast<uint16_t*>(V + M3_V_ENTRIES);
    __shared__ M3Shared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t_begin = wall_clock64();
    const int64_t cap = ctl->capacity, seg = ctl->segment_capacity;
    const int H = (int)ctl->height;
    const int64_t end = flags != nullptr ? d_T[0] : R;
    // ---- eligibility (uniform) ----
    int why = 0;
    // segments below 16 slots: a semaphore that finds the last slot as the leaf's only free one shifts cells ACROSS the leaf boundary
    // (src/writes.jl:34-38 from addpartition!, src/pcsr.jl:99-112) and the leaf count stays — typed runs on such arrays (anything
    // grown from the empty PMA keeps its 8-slot segments) are not count-only and stay with k_append_run; a vector run is
    if (!(seg == 16 || seg == 32 || (flags == nullptr && (seg == 8 || seg == 4 || seg == 2)))) why = 1;
    else if (cap < 65536 || cap > (1ll << 30) || (seg << H) != cap || H + 1 > MAX_LEVELS) why = 2;
    else if (end < M3_MIN_RUN || end >= (1ll << 30)) why = 3;
    if (why) {
        if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = why; }
        return;
    }
    // the last leaf before every insert: at least one cell (the tail is in it) and a free slot besides the last one — two free slots
    // in a typed run (the semaphore's shift must find one left of the last slot), one in a vector run
    const int maxc0 = (int)seg - (flags != nullptr ? 2 : 1);
    if (tid < 64) {
        sh.W[tid] = tid <= H ? (int32_t)(seg << tid) : 0;
        sh.lo[tid] = tid <= H ? (int32_t)ctl->lo[tid] : 1;
        sh.hi[tid] = tid <= H ? (int32_t)ctl->hi[tid] : 0;
        sh.cnt[tid] = 0; sh.ev_c[tid] = 0; sh.n[tid] = 0; sh.base[tid] = 0; sh.cmin[tid] = 0;
    }
    if (tid < 40) sh.bk[tid] = 0ull;
    if (tid == 0) { sh.H = H; sh.seg = (int32_t)seg; sh.bail = 0; sh.consumed = 0; sh.leaf_ops = 0; sh.ended = 0; sh.ev_mask = 0ull; sh.reb = 0ull; sh.slots = 0ull; sh.top_events = 0; sh.Lp = 0; }
    __syncthreads();
    m3_suffix_counts(sh, occ, cap, seg, H, tid, lane);
    const int64_t nwords = cap >> 6;
    // ---- table ranges: a count of level i never falls below the lowest suffix density of the levels >= i (minus rounding) ----
    if (tid == 0) {
        int bail = 0;
        if (sh.cnt[0] < 1 || sh.cnt[0] > maxc0) bail = 4;
        double dmin = 2.0;
        int32_t* cm = sh.tauL;                  // (scratch: the driver's records are not in use yet)
        for (int k = H; k >= 0; --k) {
            const double d = (double)sh.cnt[k] / (double)sh.W[k];
            dmin = d < dmin ? d : dmin;
            int c = (int)floor(dmin * (double)sh.W[k]) - 3;
            if (c < sh.lo[k]) c = sh.lo[k];
            if (c < 0) c = 0;
            cm[k] = c;
        }
        int Lp = 0, sum = 0;
        for (int k = 1; k < H; ++k) {
            const int nk = sh.hi[k] >= cm[k] ? sh.hi[k] - cm[k] + 1 : 0;
            if (sh.W[k] > M3_MAX_TABLE_W || nk > M3_V_ENTRIES || sum + nk > M3_X_ENTRIES) break;
            sh.cmin[k] = cm[k]; sh.n[k] = nk; sh.base[k] = sum;
            sum += nk;
            Lp = k;
        }
        if (Lp < 1) bail = 5;
        sh.Lp = Lp;
        sh.bail = bail;
    }
    __syncthreads();
    const int Lp = sh.Lp;
    if (sh.bail) {
        if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = sh.bail; }
        return;
    }
    const int64_t t_counts = wall_clock64();
    // ---- the tables, level by level: one thread per entry ----
    const int lo0 = sh.lo[0], hi0 = sh.hi[0], W0 = sh.W[0];
    for (int k = 1; k <= Lp; ++k) {
        const int nk = sh.n[k], cmk = sh.cmin[k], Wk = sh.W[k];
        for (int e = tid; e < nk; e += M3_THREADS) {
            const int c = cmk + e;
            const SpreadGeom g = make_geom(Wk, c);
            const int s0 = m3_suffix_cells(g, Wk, W0);
            bool bad = s0 < 1 || s0 > maxc0;
            int tau = m3_leaf_reject(s0, lo0, hi0);
            uint32_t reb = 0; uint64_t slots = 0;
#pragma clang loop unroll(disable)
            for (int i = 1; i < k && !bad; ++i) {
                const int ci = m3_suffix_cells(g, Wk, sh.W[i]) + tau;
                if (ci >= sh.lo[i] && ci <= sh.hi[i]) {
                    if (ci < sh.cmin[i]) { bad = true; break; }
                    const uint64_t x = X[sh.base[i] + ci - sh.cmin[i]];
                    tau += m3_dt(x); reb += m3_reb(x); slots += m3_slots(x);
                }
            }
            if (tau < 1 || tau > 0xffff || reb > 0xffffu || slots > 0xffffffffull) bad = true;
            if (bad) { sh.bail = 6; tau = 1; reb = 0; slots = 0; }
            V[e] = m3_pack((uint32_t)tau, reb, slots);
            Vdt[sh.base[k] + e] = (uint16_t)tau;
        }
        __syncthreads();
        const int hik = sh.hi[k], bk = sh.base[k];
        for (int e = tid; e < nk; e += M3_THREADS) {
            int cc = cmk + e;
            uint32_t dt = 0, reb = 0; uint64_t slots = 0;
            while (cc <= hik) {
                const uint64_t v = V[cc - cmk];
                const int dv = m3_dt(v);
                dt += (uint32_t)dv; reb += 1u + m3_reb(v); slots += (uint64_t)Wk + m3_slots(v);
                cc += dv;
            }
            if (dt > 0xffffu || reb > 0xffffu || slots > 0xffffffffull) { sh.bail = 7; dt = 1; reb = 0; slots = 0; }
            X[bk + e] = m3_pack(dt, reb, slots);
        }
        __syncthreads();
        if (sh.bail) break;
    }
    if (sh.bail) {
        if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = sh.bail; }
        return;
    }
    const int64_t t_tables = wall_clock64();
    // ---- the driver: wave 0 steps the levels above Lp event by event ----
    if (wave == 0) {
        const int end32 = (int)end, seg32 = maxc0;
        M3Lane L;
        L.W = sh.W[lane]; L.lo = sh.lo[lane]; L.hi = sh.hi[lane]; L.cmin = sh.cmin[lane]; L.base = sh.base[lane];
        int cnt = sh.cnt[lane], evc = 0, bail = 0;
        int t = 0, top = 0;
        uint64_t reb_tot = 0, slots_tot = 0, evmask = 0;
        int leaf_ops = 0, consumed = 0, ended = 0;
        M3Rec norec; norec.tau = 0; norec.reb = 0; norec.slots = 0;
        for (;;) {
            uint32_t r = 0; uint64_t s = 0;
            const int tau = m3_descent<false>(L, cnt, X, Lp + 1, r, s, bail, lane, norec);
            if (bail) break;
            if (t + tau > end32) {
                leaf_ops = m3_final_descent(L, cnt, X, Vdt, Lp + 1, end32 - t, lane, seg32, reb_tot, slots_tot, evmask, evc, bail);
                consumed = end32;
                break;
            }
            // the append t + tau visits level Lp + 1: lane <-> level evaluates the thresholds above (src/pma.jl:105-141)
            const int ck = cnt + tau;
            const bool a = lane > Lp && lane <= H && ck >= L.lo && ck <= L.hi;
            const unsigned long long am = __ballot(a);
            if (am == 0ull) {
                // no level accepts: _extend! (or _shrink!) — the run ends in front of this op
                leaf_ops = m3_final_descent(L, cnt, X, Vdt, Lp + 1, tau - 1, lane, seg32, reb_tot, slots_tot, evmask, evc, bail);
                consumed = t + tau - 1;
                ended = 1;
                break;
            }
            const int kacc = __ffsll(am) - 1;
            t += tau;
            if (lane > Lp && lane <= H) cnt = ck;
            const int c = rl(cnt, kacc);
            reb_tot += (uint64_t)r + 1ull; slots_tot += s + (uint64_t)rl(L.W, kacc);
            evmask = (evmask & ~((1ull << kacc) - 1ull)) | (1ull << kacc);
            if (lane == kacc) evc = c;
            ++top;
            if (!m3_reset_below(L, cnt, kacc, c, lane, seg32)) { bail = 8; break; }
        }
        sh.ev_c[lane] = evc;
        if (lane == 0) {
            sh.consumed = consumed; sh.leaf_ops = leaf_ops; sh.ended = ended; sh.ev_mask = evmask; sh.reb = reb_tot; sh.slots = slots_tot;
            sh.top_events = top;
            if (bail) sh.bail = bail;
        }
        wave_lds_sync();
        // ---- the last word: the narrowest surviving patterns, then the trailing leaf ops bit by bit ----
        if (lane == 0 && !sh.bail) {
            uint64_t lw = occ[nwords - 1];
            for (int k = H; k >= 1; --k) {
                if (!((evmask >> k) & 1ull)) continue;
                const int Wk = sh.W[k];
                const SpreadGeom g = make_geom(Wk, sh.ev_c[k]);
                if (Wk >= 64) lw = spread_word_bits(g, (Wk >> 6) - 1);
                else {
                    const uint64_t m = (1ull << Wk) - 1ull;
                    lw = (lw & ~(m << (64 - Wk))) | ((spread_word_bits(g, 0) & m) << (64 - Wk));
                }
            }
            const uint64_t leaf_mask = seg == 64 ? ~0ull : (((1ull << seg) - 1ull) << (64 - seg));
            constexpr uint64_t TOP = 1ull << 63;
            int bad = 0;
            for (int j = consumed - leaf_ops; j < consumed; ++j) {
                const bool is_sem = flags != nullptr && ((flags[j >> 6] >> (j & 63)) & 1ull);
                const uint64_t lf = lw & leaf_mask;
                if (lf == 0ull) { bad = 1; break; }
                const int lb = 63 - __clzll((long long)lf);                     // tail
                if (lb < 63 && !is_sem) lw |= 1ull << (lb + 1);                   // _insert! behind the tail  src/writes.jl:29-33
                else {
                    const uint64_t z = ~lw & leaf_mask & ~TOP;                    // empty slots of the leaf left of the last slot
                    if (z == 0ull) { bad = 1; break; }
                    const int pe = 63 - __clzll((long long)z);
                    if (lw & TOP) lw |= 1ull << pe;                               // cells (pe, cap] shift left  src/writes.jl:34-38
                    else if (pe == 62) lw |= TOP;
                    else { lw |= 1ull << pe; lw &= ~(1ull << 62); lw |= TOP; }
                }
            }
            if (bad) sh.bail = 9;
            sh.last_word = lw;
        }
    }
    __syncthreads();
    if (sh.bail) {
        if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = sh.bail; }
        return;
    }
    const int64_t t_driver = wall_clock64();
    // ---- commit: the surviving rebalance of every level, each up to where the next narrower one (or the last word) takes over ----
    {
        const unsigned long long evmask = sh.ev_mask;
        int64_t w_low = 64;                                      // slots at the end that a narrower pattern / the last word covers
        for (int k = 1; k <= H; ++k) {
            if (!((evmask >> k) & 1ull)) continue;
            const int64_t Wk = sh.W[k];
            if (Wk <= 64) continue;
            const SpreadGeom g = make_geom(Wk, sh.ev_c[k]);
            const int64_t w0 = (cap - Wk) >> 6, nw = (Wk - w_low) >> 6;
            for (int64_t tw = tid; tw < nw; tw += M3_THREADS) occ[w0 + tw] = spread_word_bits(g, (int)tw);
            w_low = Wk;
        }
        if (tid == 0) {
            occ[nwords - 1] = sh.last_word;
            ctl->nb_elements += sh.consumed;
            ctl->stat_rebalances += (int64_t)sh.reb; ctl->stat_window_slots += (int64_t)sh.slots; ctl->stat_small_rebalances += (int64_t)sh.reb;
            out[0] = sh.consumed; out[1] = sh.ended ? 2 : 1; out[2] = 0;
            out[3] = sh.top_events; out[4] = Lp; out[5] = t_counts - t_begin; out[6] = t_tables - t_counts; out[7] = t_driver - t_tables;
        }
    }
}


// =====================================================================================================================================
// Model v5: TYPED append runs (semaphore cells) on 8-slot segments — BASELINE config 5's colmajor orientation: a matrix grown from the
// empty one keeps its 8-slot segments for life (_extend! doubles the segment COUNT, src/pma.jl:143-151).
//
// Why the count-only tables above do not apply.  The leaf threshold of an 8-slot segment leaves ONE free slot (hi[0] = 7).  When that
// slot is the last one of the array and the next cell is a semaphore — which always goes to the last slot, with a shift to the left
// (addpartition!, src/pcsr.jl:99-112 -> _insert!, src/writes.jl:34-38) — the shift finds its gap OUTSIDE the leaf: one cell crosses the
// leaf boundary, the leaf keeps its 7 cells and accepts, no level is scanned ("cross" below).  So the time a leaf epoch takes depends on
// the cell types, and so does everything above it.  What stays true (tests/repro/model5_proto.py, checked bit for bit against the
// oracle on columns of 1..20 cells): the state is the suffix cell count of every level plus the leaf state (s cells, g free slots behind
// the tail); an epoch is the ops until the leaf holds 8 cells — 8 - s of them, one more when all free slots are behind the tail
// (s + g == 8) and the op that meets the leaf at 7 cells is a semaphore —; it ends with the rebalance of the first level >= 1 that
// accepts its count (every level below is full), which resets counts and leaf state to the closed form of spread!; a cross op adds a
// cell to every suffix window that holds the gap it fills (level >= kstar) and to none below.
//
// At the densities of an append run the suffix sits at its thresholds: 12 800 epochs per 17 000 cells, spread over ALL levels
// (13 / 21 / 26 / 20 / 8 / 5 / 3 % at levels 1..7).  Tables over (level, count) do not skip them — the epoch sequence IS the process —
// so this kernel makes the epoch cheap instead: ONE wave, lane <-> level, every loop-carried value in registers, ~45 instructions and
// one LDS read per epoch and no divergent branch on the common path:
//   * counts: one v_add; the accepting level: two compares + a ballot + s_ff1; its count: one v_readlane;
//   * the reset below the accepting level: ONE ds_read_u16 per lane from a table row R[k][c] = (s | g << 8, count of level 1, ...,
//     count of level k - 1) — rows for every level whose windows fit the table (<= 4096 slots: 99.9 % of the events), filled by the
//     whole workgroup in front of the loop (a row = one fp64 division + k closed-form counts); wider events compute their row by lanes;
//   * the semaphores of the run as a list of cell indices in LDS (one read per semaphore, issued an epoch ahead).
// Nothing is committed before the loop has ended: the surviving rebalance of every level (the last one no wider one followed) is written
// as its spread! pattern, widest first, like the count-only model.  The trailing partial epoch — and an epoch that ends in _extend! —
// is NOT consumed: k_append_run replays those < 9 cells on the bitmap.
constexpr int M5_TAB_U16 = 56 * 1024;            // 112 KB of table entries
constexpr int M5_MAX_SEMS = 8192;                // 32 KB: semaphore cells of the run (a longer run is consumed up to that one)
constexpr int64_t M5_MIN_RUN = 64;

__device__ __forceinline__ uint32_t m5_block_excl_scan(uint32_t v, uint32_t* wsum, int tid, int lane, int wave, uint32_t* total) {
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int k = 0; k < M3_THREADS / 64; ++k) { const uint32_t w = wsum[k]; if (k < wave) woff += w; tot += w; }
    __syncthreads();
    *total = tot;
    return woff + x - v;
}

__global__ __launch_bounds__(M3_THREADS) void k_append_model5(uint64_t* occ, Ctl* ctl, int64_t R, const uint64_t* flags, const int64_t* d_T,
                                                              int64_t* out, const int use_asm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char m3_lds[];
    uint16_t* Tab = reinterpret_cast<uint16_t*>(m3_lds);
    uint32_t* Sem = reinterpret_cast<uint32_t*>(Tab + M5_TAB_U16);
    __shared__ M3Shared sh;
    __shared__ uint32_t sWsum[M3_THREADS / 64];
    __shared__ int32_t sOff[64];                 // rowbase[k] - lo[k] * k (u16 units): the row of (k, c) starts at sOff[k] + c * k
    __shared__ int32_t sNSem, sKT, sEndEff;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t_begin = wall_clock64();
    const int64_t cap = ctl->capacity, seg = ctl->segment_capacity;
    const int H = (int)ctl->height;
    const int64_t end = flags != nullptr ? d_T[0] : R;
    int why = 0;
    if (flags == nullptr || seg != 8 || ctl->lo[0] != 1 || ctl->hi[0] != 7) why = 1;
    else if (cap < 256 || cap > (1ll << 30) || (seg << H) != cap || H + 1 > MAX_LEVELS || H < 2) why = 2;
    else if (end < M5_MIN_RUN || end >= (1ll << 30)) why = 3;
    if (why) {
        if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = why; }
        return;
    }
    if (tid < 64) {
        sh.W[tid] = tid <= H ? (int32_t)(seg << tid) : 0;
        sh.lo[tid] = tid <= H ? (int32_t)ctl->lo[tid] : 1;
        sh.hi[tid] = tid <= H ? (int32_t)ctl->hi[tid] : 0;
        sh.cnt[tid] = 0; sh.ev_c[tid] = 0; sh.n[tid] = 0; sh.base[tid] = 0; sh.cmin[tid] = 0; sOff[tid] = 0;
    }
    if (tid < 40) sh.bk[tid] = 0ull;
    if (tid == 0) { sh.H = H; sh.seg = 8; sh.bail = 0; sh.consumed = 0; sh.leaf_ops = 0; sh.ended = 0; sh.ev_mask = 0ull; sh.reb = 0ull; sh.slots = 0ull; sh.top_events = 0; sh.Lp = 0; }
    __syncthreads();
    m3_suffix_counts(sh, occ, cap, seg, H, tid, lane);
    const int64_t nwords = cap >> 6;
    // ---- which levels get table rows: a row of level k = k entries, one row per accepted count ----
    if (tid == 0) {
        int KT = 0, sum = 0;
        for (int k = 1; k <= H; ++k) {
            const int nk = sh.hi[k] >= sh.lo[k] ? sh.hi[k] - sh.lo[k] + 1 : 0;
            if (sum + nk * k > M5_TAB_U16) break;
            sh.base[k] = sum; sh.n[k] = nk; sOff[k] = sum - sh.lo[k] * k;
            sum += nk * k;
            KT = k;
        }
        sKT = KT;
        if (sh.cnt[0] < 1 || sh.cnt[0] > 7) sh.bail = 4;         // the tail is not in the last leaf (or the leaf is full): not this model's run
    }
    __syncthreads();
    if (sh.bail) {
        if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = sh.bail; }
        return;
    }
    const int KT = sKT;
    const int64_t t_counts = wall_clock64();
    // ---- the table rows: one thread per row — make_geom (the one fp64 division), the leaf state, the counts of the levels below ----
    for (int k = 1; k <= KT; ++k) {
        const int nk = sh.n[k], lok = sh.lo[k], Wk = sh.W[k], bk = sh.base[k];
        for (int r = tid; r < nk; r += M3_THREADS) {
            const int c = lok + r;
            const SpreadGeom g = make_geom(Wk, c);
            const int s0 = m3_suffix_cells(g, Wk, 8);
            const int g0 = (int)(Wk - spread_last_cell(g));
            uint16_t* row = Tab + bk + r * k;
            row[0] = (uint16_t)((8 - s0) | ((s0 + g0 == 8 ? 8 - s0 : 0) << 4));      // the leaf state as the epoch loop reads it (see e0 there)
            if (s0 < 1 || s0 > 7 || g0 < 0 || g0 > 8 - s0) sh.bail = 6;         // (cannot happen: the last gap of spread! lies on one of the last two slots)
            for (int i = 1; i < k; ++i) row[i] = (uint16_t)m3_suffix_cells(g, Wk, sh.W[i]);
        }
    }
    // ---- the semaphore cells of the run: their cell indices, ascending, as a list ----
    {
        const int64_t fwords = (end + 63) >> 6;
        uint32_t base_n = 0;
        int32_t end_eff = (int32_t)end;
        for (int64_t w0 = 0; w0 < fwords; w0 += M3_THREADS) {
            const int64_t w = w0 + tid;
            uint64_t f = w < fwords ? flags[w] : 0ull;
            if (w < fwords && ((w + 1) << 6) > end) f &= mask_lt((int)(end - (w << 6)));
            uint32_t tot = 0;
            const uint32_t at = base_n + m5_block_excl_scan((uint32_t)popc64(f), sWsum, tid, lane, wave, &tot);
            uint32_t q = at;
            while (f) {
                const int b = __ffsll((unsigned long long)f) - 1;
                f &= f - 1;
                if (q < (uint32_t)M5_MAX_SEMS) Sem[q] = (uint32_t)((w << 6) + b);
                ++q;
            }
            base_n += tot;
            if (base_n > (uint32_t)M5_MAX_SEMS) break;            // (uniform: base_n is the same in every thread)
        }
        __syncthreads();
        if (tid == 0) {
            // more semaphores than the list holds: the run is consumed up to the first one that did not fit
            if (base_n > (uint32_t)M5_MAX_SEMS) { end_eff = (int32_t)Sem[M5_MAX_SEMS - 1]; base_n = M5_MAX_SEMS - 1; }
            sNSem = (int32_t)base_n; sEndEff = end_eff;
        }
    }
    __syncthreads();
    if (sh.bail) {
        if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = sh.bail; }
        return;
    }
    const int64_t t_tables = wall_clock64();
    // ---- the epochs: wave 0, lane <-> level ----
    if (wave == 0) {
        const int end32 = ufl(sEndEff), nsem = ufl(sNSem), KTu = ufl(KT);
        // per-lane constants of the level; lanes that are no level (0, > H) never accept and never hold a gap
        const bool lvl = lane >= 1 && lane <= H;
        const int lol = lvl ? sh.lo[lane] : 0x7fffffff;
        const unsigned rngl = lvl ? (unsigned)(sh.hi[lane] - sh.lo[lane]) : 0u;
        const int Wm1 = lvl ? sh.W[lane] - 1 : (int)0x80000000;
        const int Wl = sh.W[lane];
        const int off2 = 2 * sOff[lane] + (int)(reinterpret_cast<uintptr_t>(Tab) & 0xffffffffu);      // byte address of "row 0" of the level (LDS)
        const int lane2 = 2 * lane;
        int cnt = sh.cnt[lane], evc = 0, evt = -1;
        unsigned nev = 0;                                  // events of the level (lane <-> level): rebalances and window slots at the end
        // leaf state as the table rows carry it: e0 = (8 - s) | (jx + 1) << 4, jx = 7 - s when every free slot is behind the tail (a semaphore
        // that meets the leaf at 7 cells then shifts across the leaf boundary), jx + 1 = 0 otherwise
        int e0;
        {
            const int s0 = sh.cnt[0];
            const uint32_t leafbits = (uint32_t)(occ[nwords - 1] >> 56);
            const int g0 = __clz((int)leafbits) - 24;
            e0 = ufl((8 - s0) | ((s0 + g0 == 8 ? 8 - s0 : 0) << 4));
        }
        int t = 0, sp = 0, bail = 0;
        constexpr int NOSEM = 0x7fffffff;
        const int semb = ufl((int)(reinterpret_cast<uintptr_t>(Sem) & 0xffffffffu));      // LDS byte address of the semaphore list
        int ns = nsem > 0 ? ufl((int)Sem[0]) : NOSEM;
        int ns_next = nsem > 1 ? ufl((int)Sem[1]) : NOSEM;
        // The common epoch — no cross-leaf shift, a level with a table row accepts, the next semaphore still ahead — is a hand-scheduled
        // block of 36 instructions (the compiler's version of the same loop: 55, with its exits merged through mask registers; measured
        // 165 ns per epoch against ~90): one LDS read per epoch, issued as early as its address is known, the bookkeeping of the
        // rebalanced level under its latency.  Everything else leaves the block with a status and takes ONE epoch of the generic C++
        // path below: 1 cross-leaf shift, 2 event above the tables; 0 = the run's end or _extend!.  The next semaphore is taken from the list inside the block.
        // Registers: lane <-> level in cnt / evc / evt / nev; e0, t, ns wave-uniform in SGPRs.  Hazards (gfx950): lane selects of
        // v_readlane come from SALU results, every VALU-written SGPR is consumed by SALU or after >= 2 instructions.
        for (;;) {
            // the next semaphore at or behind cell t (semaphores are at least two cells apart — a new column comes with its first
            // element —, epochs at most 9 cells long: a step or two)
            while (ns < t) {
                ++sp;
                ns = ns_next;
                ns_next = sp + 1 < nsem ? ufl((int)Sem[sp + 1]) : NOSEM;
            }
            int status = 4;
            if (use_asm) {
                int r_ln, r_tmp, r_tn, r_k, r_c, r_ra;
                int v_cn, v_t, v_nv;
                asm volatile(
                    "L_m5_top_%=:\n\t"
                    "s_and_b32 %[ln], %[e0], 15\n\t"
                    "s_lshr_b32 %[tmp], %[e0], 4\n\t"
                    "s_sub_i32 %[tn], %[ns], %[t]\n\t"
                    "s_add_i32 %[tn], %[tn], 1\n\t"
                    "s_cmp_eq_u32 %[tn], %[tmp]\n\t"
                    "s_cbranch_scc1 L_m5_x1_%=\n\t"
                    "s_add_i32 %[tn], %[t], %[ln]\n\t"
                    "s_cmp_gt_i32 %[tn], %[end]\n\t"
                    "s_cbranch_scc1 L_m5_x0_%=\n\t"
                    "v_add_u32_e32 %[cn], %[ln], %[cnt]\n\t"
                    "v_sub_u32_e32 %[vt], %[cn], %[lol]\n\t"
                    "v_cmp_le_u32_e32 vcc, %[vt], %[rng]\n\t"
                    "s_cbranch_vccz L_m5_x0_%=\n\t"
                    "s_ff1_i32_b64 %[k], vcc\n\t"
                    "s_cmp_gt_i32 %[k], %[kt]\n\t"
                    "s_cbranch_scc1 L_m5_x2_%=\n\t"
                    "v_readlane_b32 %[c], %[cn], %[k]\n\t"
                    "v_readlane_b32 %[ra], %[off2], %[k]\n\t"
                    "s_lshl_b32 %[tmp], %[k], 1\n\t"
                    "s_mul_i32 %[tmp], %[tmp], %[c]\n\t"
                    "s_add_i32 %[ra], %[ra], %[tmp]\n\t"
                    "v_add_u32_e32 %[vt], %[ra], %[lane2]\n\t"
                    "ds_read_u16 %[nv], %[vt]\n\t"
                    "v_cmp_eq_u32_e32 vcc, %[k], %[lane]\n\t"
                    "v_mov_b32_e32 %[vt], %[c]\n\t"
                    "v_cndmask_b32_e32 %[evc], %[evc], %[vt], vcc\n\t"
                    "v_mov_b32_e32 %[vt], %[tn]\n\t"
                    "v_cndmask_b32_e32 %[evt], %[evt], %[vt], vcc\n\t"
                    "v_addc_co_u32_e32 %[nev], vcc, 0, %[nev], vcc\n\t"
                    "v_cmp_gt_u32_e32 vcc, %[k], %[lane]\n\t"
                    "s_mov_b32 %[t], %[tn]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_cndmask_b32_e32 %[cnt], %[cn], %[nv], vcc\n\t"
                    "v_readfirstlane_b32 %[e0], %[nv]\n\t"
                    "s_cmp_lt_i32 %[ns], %[t]\n\t"
                    "s_cbranch_scc0 L_m5_top_%=\n"
                    // the semaphore has been passed: the next one from the list (one LDS read per semaphore, one in 13 epochs)
                    "L_m5_adv_%=:\n\t"
                    "s_mov_b32 %[ns], %[nsn]\n\t"
                    "s_add_i32 %[sp], %[sp], 1\n\t"
                    "s_add_i32 %[tmp], %[sp], 1\n\t"
                    "s_mov_b32 %[nsn], 0x7fffffff\n\t"
                    "s_cmp_lt_i32 %[tmp], %[nsem]\n\t"
                    "s_cbranch_scc0 L_m5_chk_%=\n\t"
                    "s_lshl_b32 %[tmp], %[tmp], 2\n\t"
                    "s_add_i32 %[tmp], %[tmp], %[semb]\n\t"
                    "v_mov_b32_e32 %[vt], %[tmp]\n\t"
                    "ds_read_b32 %[vt], %[vt]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_readfirstlane_b32 %[nsn], %[vt]\n"
                    "L_m5_chk_%=:\n\t"
                    "s_cmp_lt_i32 %[ns], %[t]\n\t"
                    "s_cbranch_scc1 L_m5_adv_%=\n\t"
                    "s_branch L_m5_top_%=\n"
                    "L_m5_x0_%=:\n\t"
                    "s_mov_b32 %[st], 0\n\t"
                    "s_branch L_m5_out_%=\n"
                    "L_m5_x1_%=:\n\t"
                    "s_mov_b32 %[st], 1\n\t"
                    "s_branch L_m5_out_%=\n"
                    "L_m5_x2_%=:\n\t"
                    "s_mov_b32 %[st], 2\n"
                    "L_m5_out_%=:\n\t"
                    : [st] "=&s"(status), [e0] "+s"(e0), [t] "+s"(t), [ns] "+s"(ns), [nsn] "+s"(ns_next), [sp] "+s"(sp), [cnt] "+v"(cnt), [evc] "+v"(evc), [evt] "+v"(evt), [nev] "+v"(nev),
                      [ln] "=&s"(r_ln), [tmp] "=&s"(r_tmp), [tn] "=&s"(r_tn), [k] "=&s"(r_k), [c] "=&s"(r_c), [ra] "=&s"(r_ra),
                      [cn] "=&v"(v_cn), [vt] "=&v"(v_t), [nv] "=&v"(v_nv)
                    : [nsem] "s"(nsem), [semb] "s"(semb), [end] "s"(end32), [kt] "s"(KTu), [lol] "v"(lol), [rng] "v"(rngl), [off2] "v"(off2), [lane2] "v"(lane2), [lane] "v"(lane)
                    : "vcc", "scc", "memory");
                if (status == 0) break;                  // (the generic path would break at the same test: the trailing epoch / _extend! is k_append_run's)
            }
            // ---- one epoch, every case ----
            const int ln0 = e0 & 15, jxp1 = e0 >> 4;
            const bool cross = ns - t + 1 == jxp1;                   // (jx + 1 = 0: never — ns >= t)
            const int ln = ln0 + (cross ? 1 : 0);
            const int tn = t + ln;
            if (tn > end32) break;                                   // the trailing partial epoch: k_append_run's
            int cn = cnt + ln;
            if (__builtin_expect(cross, 0)) {
                // the gap the shift fills: the first level whose suffix window has a free slot besides the last one; the windows below
                // it lose the cell that crosses their left boundary for the one that comes in
                const unsigned long long fm = __builtin_amdgcn_ballot_w64(cnt + (jxp1 - 1) < Wm1);
                if (fm == 0ull) { bail = 8; break; }
                const int kstar = __ffsll(fm) - 1;
                if (lane < kstar) cn -= 1;
            }
            const unsigned long long am = __builtin_amdgcn_ballot_w64((unsigned)(cn - lol) <= rngl);
            if (am == 0ull) break;                                   // no level accepts: _extend! — the epoch is k_append_run's too
            const int kacc = __ffsll(am) - 1;
            const int c = rl(cn, kacc);
            int nv;
            if (__builtin_expect(kacc <= KTu, 1)) {
                const int rowaddr = rl(off2, kacc) + c * (kacc << 1);
                nv = (int)*reinterpret_cast<const __attribute__((address_space(3))) uint16_t*>((uintptr_t)(unsigned)(rowaddr + lane2));
            } else {
                // an event above the tables (windows beyond 4096 slots: one in a thousand): its row by lanes
                const int Wk = rl(Wl, kacc);
                const SpreadGeom gg = make_geom(Wk, c);
                const int s1 = m3_suffix_cells(gg, Wk, 8), g1 = (int)(Wk - spread_last_cell(gg));
                nv = lane == 0 ? ((8 - s1) | ((s1 + g1 == 8 ? 8 - s1 : 0) << 4)) : (lane < kacc ? m3_suffix_cells(gg, Wk, Wl) : 0);
            }
            t = tn;
            {   // bookkeeping of the level that was rebalanced (independent of the table read: issued under its latency)
                const bool me = lane == kacc;
                evc = me ? c : evc;
                evt = me ? tn : evt;
                nev += me ? 1u : 0u;
            }
            cnt = lane < kacc ? nv : cn;
            e0 = ufl(nv);
        }
        // rebalances and window slots: per level, summed over the lanes
        unsigned long long slots = (unsigned long long)nev * (unsigned long long)(lvl ? Wl : 0);
        unsigned epochs = nev;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { slots += __shfl_xor(slots, o, 64); epochs += __shfl_xor(epochs, o, 64); }
        sh.ev_c[lane] = evc;
        // the surviving rebalance of every level: no wider one came later
        int later = -1;
        for (int k = H; k >= 1; --k) {
            const int e = rl(evt, k);
            if (lane == 0 && e > later) sh.ev_mask |= 1ull << k;
            later = e > later ? e : later;
        }
        if (lane == 0) {
            sh.consumed = t; sh.leaf_ops = 0; sh.ended = 0; sh.reb = (unsigned long long)epochs; sh.slots = slots; sh.top_events = epochs;
            if (bail) sh.bail = bail;
        }
        wave_lds_sync();
        // ---- the last word: the narrowest surviving patterns (no trailing ops: they were not consumed) ----
        if (lane == 0 && !sh.bail) {
            const unsigned long long evmask = sh.ev_mask;
            uint64_t lw = occ[nwords - 1];
            for (int k = H; k >= 1; --k) {
                if (!((evmask >> k) & 1ull)) continue;
                const int Wk = sh.W[k];
                const SpreadGeom gk = make_geom(Wk, sh.ev_c[k]);
                if (Wk >= 64) lw = spread_word_bits(gk, (Wk >> 6) - 1);
                else {
                    const uint64_t m = (1ull << Wk) - 1ull;
                    lw = (lw & ~(m << (64 - Wk))) | ((spread_word_bits(gk, 0) & m) << (64 - Wk));
                }
            }
            sh.last_word = lw;
        }
    }
    __syncthreads();
    if (sh.bail || sh.consumed == 0) {
        if (tid == 0) { out[0] = 0; out[1] = 0; out[2] = sh.bail ? sh.bail : 9; }
        return;
    }
    const int64_t t_driver = wall_clock64();
    // ---- commit: the surviving rebalance of every level, each up to where the next narrower one (or the last word) takes over ----
    {
        const unsigned long long evmask = sh.ev_mask;
        int64_t w_low = 64;
        for (int k = 1; k <= H; ++k) {
            if (!((evmask >> k) & 1ull)) continue;
            const int64_t Wk = sh.W[k];
            if (Wk <= 64) continue;
            const SpreadGeom g = make_geom(Wk, sh.ev_c[k]);
            const int64_t w0 = (cap - Wk) >> 6, nw = (Wk - w_low) >> 6;
            for (int64_t tw = tid; tw < nw; tw += M3_THREADS) occ[w0 + tw] = spread_word_bits(g, (int)tw);
            w_low = Wk;
        }
        if (tid == 0) {
            occ[nwords - 1] = sh.last_word;
            ctl->nb_elements += sh.consumed;
            ctl->stat_rebalances += (int64_t)sh.reb; ctl->stat_window_slots += (int64_t)sh.slots; ctl->stat_small_rebalances += (int64_t)sh.reb;
            out[0] = sh.consumed; out[1] = 1; out[2] = 0;
            out[3] = sh.top_events; out[4] = KT; out[5] = t_counts - t_begin; out[6] = t_tables - t_counts; out[7] = t_driver - t_tables;
        }
    }
}


}  // namespace

hipError_t launch_append_model3(uint64_t* occ, Ctl* ctl, int64_t R, const uint64_t* flags, const int64_t* d_T, int64_t* out, hipStream_t stream) {
    constexpr size_t LDS = (size_t)(M3_X_ENTRIES + M3_V_ENTRIES) * sizeof(uint64_t) + (size_t)M3_X_ENTRIES * sizeof(uint16_t);
    static PerDeviceOnce once;
    {
        hipError_t e = once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(k_append_model3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_append_model3, dim3(1), dim3(M3_THREADS), LDS, stream, occ, ctl, R, flags, d_T, out);
    return hipGetLastError();
}


hipError_t launch_append_model5(uint64_t* occ, Ctl* ctl, int64_t R, const uint64_t* flags, const int64_t* d_T, int64_t* out, hipStream_t stream) {
    constexpr size_t LDS = (size_t)M5_TAB_U16 * sizeof(uint16_t) + (size_t)M5_MAX_SEMS * sizeof(uint32_t);
    static PerDeviceOnce once;
    {
        hipError_t e = once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(k_append_model5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        });
        if (e != hipSuccess) return e;
    }
    // dev knob DSA_MODEL5=2: every epoch through the generic C++ path (A/B against the hand-scheduled block, coverage of that path)
    static const int use_asm = [] { const char* e = dev_env("DSA_MODEL5"); return (e && e[0] == '2') ? 0 : 1; }();
    hipLaunchKernelGGL(k_append_model5, dim3(1), dim3(M3_THREADS), LDS, stream, occ, ctl, R, flags, d_T, out, use_asm);
    return hipGetLastError();
}

}  // namespace dsa
