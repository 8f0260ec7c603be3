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


}  // namespace dsa
