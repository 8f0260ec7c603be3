// csrc/rebalance.hip — K-pack + K-spread fused: the grid-wide window rebalance for gfx950.
//
// Reproduces  pack!  (src/moves.jl:94-110)  followed by  spread!  (src/moves.jl:120-171) of the
// reference — i.e. _even_rebalance! (src/pma.jl:94-103, src/pcsr.jl:88-97), the full spread of
// _pma (src/pma.jl:42-55) and the pack + _shrink! / _extend! paths (src/pma.jl:135-161) — as ONE
// out-of-place pass: the r-th occupied source cell goes straight to the r-th non-gap offset of the
// destination window, whose gaps sit at the closed-form offsets floor(fl(k*fl(W/E))) (dsa_dev.h).
//
// Bound: HBM.  Algorithmic bytes per W-slot window = 2 * 16 * W (read + write every slot);
// the occupancy bitmap (W/8 B each way) and the tile counters are not counted.
//
//   k_move2      : the whole rebalance in ONE launch, one workgroup per 2048-slot SOURCE tile (see the comment at the
//                  kernel): prefix table published by the first workgroups, cells compacted into LDS by in-tile rank,
//                  every owned destination offset — gaps included, so lines are written whole — stored as 16-byte
//                  key / value pairs, occupancy words closed-form, semaphore positions scattered to the table.
//   k_tile_count / k_tile_scan : flat prefix (count, then a one-workgroup scan) for k_compact / k_permute / views
#include "dsa_dev.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace dsa {

constexpr int SRC_TILE_WORDS = 64;                 // 4096 slots
constexpr int64_t SRC_TILE = 64 * SRC_TILE_WORDS;
constexpr int MOVE_BLOCK = 256;

__device__ __forceinline__ uint64_t range_mask_for_word(int64_t w, int64_t lo0, int64_t hi0) {
    // bits of word w (slots 64w .. 64w+63, 0-based) that lie inside [lo0, hi0] (0-based, inclusive)
    const int64_t b = w << 6;
    int64_t a = lo0 - b, z = hi0 - b;
    if (z < 0 || a > 63) return 0ull;
    if (a < 0) a = 0;
    if (z > 63) z = 63;
    const uint64_t upto = (z == 63) ? ~0ull : ((1ull << (z + 1)) - 1ull);
    return upto & ~mask_lt((int)a);
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v) {
    const int lane = lane_id();
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x - v;
}

// 256-thread workgroups, 16 tiles each (wave w: tiles 16b+4w .. +3): the four occupancy loads of a lane are in flight together
constexpr int CNT_TILES = 16;
__global__ __launch_bounds__(256) void k_tile_count(const uint64_t* __restrict__ occ, int64_t lo0, int64_t hi0,
                                                    int64_t w0, int64_t nwords, int64_t ntiles, uint32_t* __restrict__ tile_cnt) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t pc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = (int64_t)blockIdx.x * CNT_TILES + wv * 4 + i;
        const int64_t w = t * SRC_TILE_WORDS + lane;
        pc[i] = 0;
        if (t < ntiles && w < nwords) pc[i] = popc64(occ[w0 + w] & range_mask_for_word(w0 + w, lo0, hi0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = (int64_t)blockIdx.x * CNT_TILES + wv * 4 + i;
        const uint32_t r = wave_reduce_add(pc[i]);
        if (lane == 0 && t < ntiles) tile_cnt[t] = r;
    }
}

// the same for TWO bitmaps in one launch (K-permute needs the tile prefixes of the old and of the new bitmap): workgroups [0, nbA)
// count A, the rest B
__global__ __launch_bounds__(256) void k_tile_count2(const uint64_t* __restrict__ occA, int64_t hiA, int64_t nwordsA, int64_t ntilesA,
                                                     uint32_t* __restrict__ cntA, int nbA,
                                                     const uint64_t* __restrict__ occB, int64_t hiB, int64_t nwordsB, int64_t ntilesB,
                                                     uint32_t* __restrict__ cntB) {
    const bool second = (int)blockIdx.x >= nbA;
    const uint64_t* __restrict__ occ = second ? occB : occA;
    const int64_t hi0 = second ? hiB : hiA, nwords = second ? nwordsB : nwordsA, ntiles = second ? ntilesB : ntilesA;
    uint32_t* __restrict__ tile_cnt = second ? cntB : cntA;
    const int64_t blk = second ? (int64_t)blockIdx.x - nbA : (int64_t)blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t pc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = blk * CNT_TILES + wv * 4 + i;
        const int64_t w = t * SRC_TILE_WORDS + lane;
        pc[i] = 0;
        if (t < ntiles && w < nwords) pc[i] = popc64(occ[w] & range_mask_for_word(w, 0, hi0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = blk * CNT_TILES + wv * 4 + i;
        const uint32_t r = wave_reduce_add(pc[i]);
        if (lane == 0 && t < ntiles) tile_cnt[t] = r;
    }
}

// one workgroup: exclusive prefix of n counts, 4096 per round (4 consecutive counts per thread, one barrier pair per round); a second
// workgroup (blockIdx 1) scans a second array in the same launch
__global__ __launch_bounds__(1024) void k_tile_scan(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off, int64_t n,
                                                    const uint32_t* __restrict__ cnt2 = nullptr, uint32_t* __restrict__ off2 = nullptr, int64_t n2 = 0) {
    if (blockIdx.x == 1) { cnt = cnt2; off = off2; n = n2; }
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t carry = 0;
    for (int64_t base = 0; base < n; base += 4096) {
        const int64_t i0 = base + (int64_t)tid * 4;
        uint32_t c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = i0 + i < n ? cnt[i0 + i] : 0u;
        const uint32_t local = c[0] + c[1] + c[2] + c[3];
        const uint32_t ex = wave_excl_scan(local);
        if (lane == 63) wsum[wv] = ex + local;
        __syncthreads();
        uint32_t run = carry + ex, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t wsk = wsum[k]; if (k < wv) run += wsk; total += wsk; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (i0 + i < n) off[i0 + i] = run; run += c[i]; }
        carry += total;
        __syncthreads();
    }
    if (tid == 0) off[n] = carry;
}

// ---- k_move2: the window rebalance in ONE launch (source-driven, chained scan) --------------------------------------------
// One 256-thread workgroup per 2048-slot SOURCE tile, in blockIdx order:
//   1. the 32 occupancy words of the tile -> cell count c and in-tile word bases; all slot loads of the tile are issued
//      with them (lane <-> slot, occupied or not);
//   2. the cells are compacted into LDS by in-tile rank;
//   0. the first ntiles/64 workgroups — dispatched first — each count one group of 64 tiles straight from the bitmap and
//      publish the prefix table (status words = {generation, value}: exclusive prefix of every tile inside its group,
//      total of every group) before doing their own tile;
//   3. the number P of cells in front of the tile = its table entry + totals of the groups in front: ONE round of
//      agent-scope loads issued together with the slot loads.  A workgroup only ever waits for LOWER block indices,
//      which the hardware dispatched earlier and which wait for nobody; past the first wave of workgroups the table is
//      complete before a workgroup starts (a per-tile chained scan was measured at 8-10 us of waiting PER TILE at 2^24
//      slots: status words cross the fabric behind the bulk stores of the CUs that publish them);
//   4. the tile owns the destination offsets (dest(P), dest(P + c)] — its cells and the gaps in front of each of
//      them, the last tile also the gaps behind the last cell — with dest(r) = r + #gaps in front of rank r from the
//      closed-form gap positions D(k) = floor(fl(k * fl(W/E))) (dsa_dev.h; scatter recipe of SURVEY App. A.3, at most
//      one fix-up step).  Every owned offset is stored (gaps too: whole lines), occupancy words are closed-form in
//      (W, m) and written by the owner of their first offset, semaphore cells update semaphores[id].
// No count kernel, no scan kernel, no interpolation search: the time does not depend on how the cells are distributed
// over the source window (uniform after a spread, packed to the left after appends, anything in between).
// PACKED sources (K-build, pack! + _shrink!: cells are the first m slots) have closed-form prefixes and skip 1-3.
constexpr int M2_TILE_BIG = 2048;      // source slots per workgroup (M2_TILE_SMALL: dev knob, see launch_rebalance)
constexpr int M2_TILE_SMALL = 1024;

struct Move2Args {
    KeyArr src_keys; const double* src_vals; const uint64_t* src_occ;
    int64_t src_lo0, src_hi0;      // 0-based inclusive source slot range (src_lo0 a multiple of 64)
    KeyArr dst_keys; double* dst_vals; uint64_t* dst_occ;
    int64_t dst_lo0;               // 0-based first destination slot
    int64_t Wd, m;
    int64_t* sems;
    unsigned long long* status;    // [0, ntiles): cells in front of the tile inside its group of 64, [ntiles, ..): cells per group; word = [63:34] generation, [31:0] value
    unsigned long long gen;
    int64_t ntiles;
    unsigned long long* fault;     // raised by a workgroup that gave up waiting for a status word (see M2_SPIN_MAX)
    double cells_to_gaps;          // E / m: starting guess of #gaps in front of a rank
    double geom_f, geom_inv_f;     // SpreadGeom::f / inv_f of (Wd, m), divided once on the host (IEEE: the same doubles as on the device)
    int dbg;                       // DSA_DBG_MOVE2 ablation knob (dev only): 1 = no waiting for status words, 2 = no write phase, 4 = closed-form P,
                                   // 8 = the group totals are NOT published, 16 = give up after 4096 polls (8 + 16: the fault path, tests)
};

// 1-based destination offset of the cell of rank r (1 <= r <= m)
__device__ __forceinline__ int dest_of_rank(const SpreadGeom& g, int r, double cells_to_gaps) {
    const int E = (int)g.E;
    if (E <= 0) return r;
    int k = (int)((double)r * cells_to_gaps);
    if (k > E) k = E;
    if (k < 0) k = 0;
#pragma clang loop vectorize(disable) unroll(disable)
    while (k < E && gap_D(g, k + 1) <= r + k) ++k;
#pragma clang loop vectorize(disable) unroll(disable)
    while (k > 0 && gap_D(g, k) >= r + k) --k;
    return r + k;
}

// DISPATCH-ORDER ASSUMPTION.  Workgroup t only ever waits for status words published by workgroups with a LOWER blockIdx (the first
// ntiles/64 of the same launch), which wait for nobody.  That is deadlock-free as long as the hardware starts workgroups in blockIdx
// order — what every AMD command processor does for a 1-D grid — so that a waiting workgroup can never occupy the slot its producer
// needs.  The wait is nevertheless bounded: after M2_SPIN_MAX polls (~seconds) a workgroup raises Move2Args::fault and carries on
// with whatever it read (the result is garbage, the stream is not hung); the host turns the flag into DSA_EHIP at its next
// synchronisation point (pma_sync_check) and the invariant checker reports it.
constexpr unsigned int M2_SPIN_MAX = 1u << 24;

template <bool PACKED, bool WIDE, int BLOCK, int TILE>
__global__ __launch_bounds__(BLOCK) void k_move2(Move2Args a) {
    constexpr int M2_TILE = TILE, M2_WORDS = TILE / 64;     // source slots / occupancy words per workgroup
    constexpr int NW = BLOCK / 64;            // waves per workgroup
    typedef typename std::conditional<WIDE, int64_t, int32_t>::type key_t;
    const key_t* __restrict__ srck = static_cast<const key_t*>(a.src_keys.p);
    key_t* __restrict__ dstk = static_cast<key_t*>(a.dst_keys.p);
    __shared__ key_t sK[M2_TILE];
    __shared__ double sV[M2_TILE];
    __shared__ uint32_t sSum[NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t t = blockIdx.x;
    SpreadGeom g;
    g.W = a.Wd; g.E = a.Wd - a.m; g.f = a.geom_f; g.inv_f = a.geom_inv_f;
    int64_t P;      // cells in front of this tile
    int c;          // cells of this tile
    if (PACKED) {
        P = t * M2_TILE < a.m ? t * M2_TILE : a.m;
        c = (int)(a.m - P < M2_TILE ? a.m - P : M2_TILE);
        // the tile's cells, staged through LDS like those of a general source: ONE round of coalesced loads per thread (round 4 read
        // each destination pair's cells from global memory inside the write loop: one dependent round trip per 128 offsets and wave —
        // 148 us for the 10 M cells of config 3's build, twice the time of the general source that reads 70 % more)
        constexpr int CPT = M2_TILE / BLOCK;
        const key_t* __restrict__ kp = srck + a.src_lo0 + P;
        const double* __restrict__ vp = a.src_vals + a.src_lo0 + P;
        key_t kk[CPT]; double vv[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int i = tid + u * BLOCK;
            const int ic = i < c ? i : (c > 0 ? c - 1 : 0);
            kk[u] = __builtin_nontemporal_load(kp + ic);
            vv[u] = __builtin_nontemporal_load(vp + ic);
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) { const int i = tid + u * BLOCK; sK[i] = kk[u]; sV[i] = vv[u]; }
        __syncthreads();
    } else {
        const int64_t w0 = (a.src_lo0 >> 6) + t * M2_WORDS;
        const int64_t wlast = a.src_hi0 >> 6;
        const int64_t ngroups = (a.ntiles + 63) >> 6;
        unsigned long long* gstatus = a.status + a.ntiles;            // [ntiles, ntiles + ngroups): cells per group of 64 tiles
        // ---- 0. the first `ngroups` workgroups (dispatched first) build the prefix table every tile reads: workgroup j counts
        //         the 64 tiles of group j straight from the bitmap (2048 words, 8 per thread) and publishes the exclusive prefix
        //         of each tile inside the group and the group total.  They depend on nobody.
        if (t < ngroups) {
            const int64_t gw0 = (a.src_lo0 >> 6) + t * (64 * M2_WORDS);
            constexpr int WPT = 64 * M2_WORDS / BLOCK;               // words per thread (8 at 256 threads); NW threads per tile
            uint32_t pc = 0;
            if (gw0 + 64 * M2_WORDS - 1 < wlast) {                   // the whole group lies inside the window: no range masks (every group but the last)
                const uint64_t* __restrict__ wp = a.src_occ + gw0 + tid * WPT;
#pragma unroll
                for (int u = 0; u < WPT; ++u) pc += popc64(wp[u]);
            } else {
#pragma unroll
                for (int u = 0; u < WPT; ++u) {
                    const int64_t w = gw0 + (int64_t)tid * WPT + u;
                    if (w <= wlast) pc += popc64(a.src_occ[w] & range_mask_for_word(w, a.src_lo0, a.src_hi0));
                }
            }
#pragma unroll
            for (int o = 1; o < NW; o <<= 1) pc += __shfl_xor(pc, o, 64);      // lanes NW*i .. NW*i + NW-1 hold one tile
            __shared__ uint32_t sTile[64];
            if ((lane & (NW - 1)) == 0) sTile[tid / NW] = pc;
            __syncthreads();
            if (wv == 0) {
                const uint32_t tc = sTile[lane];
                const uint32_t ex = wave_excl_scan(tc);
                const int64_t tt = (t << 6) + lane;
                if (tt < a.ntiles) __hip_atomic_store(a.status + tt, (a.gen << 34) | (unsigned long long)ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 63 && !(a.dbg & 8)) __hip_atomic_store(gstatus + t, (a.gen << 34) | (unsigned long long)(ex + tc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- 1. ONE round of loads: the 32 occupancy words of the tile (every wave reads them: no barrier) and, without
        //         waiting for them, the slots of this wave's 8 words — lane <-> slot, occupied or not: the lines are fetched
        //         whole anyway, and the loads do not depend on the occupancy word.  16 loads in flight per lane.
        uint64_t myword = 0;
        constexpr int WPW = M2_WORDS / NW;     // 8 at 256 threads
        key_t kk[WPW]; double vv[WPW];
        if (w0 + M2_WORDS - 1 < wlast) {
            // a tile inside the window (every tile but the last): no range masks, no clamping, 32-bit offsets from a scalar base —
            // at 2^20 slots this kernel is bound by instruction issue (the 64-bit index arithmetic of the general form was ~150
            // instructions per wave)
            if (lane < M2_WORDS) myword = a.src_occ[w0 + lane];
            const key_t* __restrict__ kp = srck + (w0 << 6);
            const double* __restrict__ vp = a.src_vals + (w0 << 6);
            const unsigned off = (unsigned)(wv * (WPW * 64) + lane);
#pragma unroll
            for (int u = 0; u < WPW; ++u) {
                kk[u] = __builtin_nontemporal_load(kp + (off + u * 64));
                vv[u] = __builtin_nontemporal_load(vp + (off + u * 64));
            }
        } else {
            if (lane < M2_WORDS && w0 + lane <= wlast) myword = a.src_occ[w0 + lane] & range_mask_for_word(w0 + lane, a.src_lo0, a.src_hi0);
#pragma unroll
            for (int u = 0; u < WPW; ++u) {
                int64_t sidx = ((w0 + wv * WPW + u) << 6) + lane;
                if (sidx > a.src_hi0) sidx = a.src_hi0;               // a tile that ends behind the source range
                kk[u] = __builtin_nontemporal_load(srck + sidx);
                vv[u] = __builtin_nontemporal_load(a.src_vals + sidx);
            }
        }
        // ---- 2. ask for the prefix: the tile's entry (wave 0) and the totals of the groups in front (waves 1-3); in the
        //         steady state the table is long complete and the answers arrive with the slots
        const int64_t grp = t >> 6;
        const unsigned long long ready = a.gen << 34;                 // a word that counts as published and adds nothing
        const unsigned long long* poll = nullptr;
        if (!(a.dbg & 1)) {
            if (wv == 0) { if (lane == 0) poll = a.status + t; }
            else if (tid - 64 < grp) poll = gstatus + (tid - 64);
        }
        unsigned long long st0 = poll ? __hip_atomic_load(poll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ready;
        // ---- compact into LDS by in-tile rank ---------------------------------------------------------------------------------
        const uint32_t mypc = popc64(myword);
        const uint32_t myex = wave_excl_scan(mypc);
        c = (int)__builtin_amdgcn_readlane((int)(myex + mypc), M2_WORDS - 1);
        const int wv_s = __builtin_amdgcn_readfirstlane(wv);       // (wave-uniform: the word of another lane comes by v_readlane, not through the LDS crossbar)
#pragma unroll
        for (int u = 0; u < WPW; ++u) {
            const int j = wv_s * WPW + u;
            const uint64_t wm = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(myword >> 32), j) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)myword, j);
            const uint32_t wb = (uint32_t)__builtin_amdgcn_readlane((int)myex, j);
            if ((wm >> lane) & 1ull) {
                const int r = (int)wb + popc64(wm & mask_lt(lane));
                sK[r] = kk[u]; sV[r] = vv[u];
            }
        }
        // ---- 3. cells in front of the tile ------------------------------------------------------------------------------------
        const unsigned int spin_max = (a.dbg & 16) ? 4096u : M2_SPIN_MAX;
        for (unsigned int spin = 0; (st0 >> 34) != a.gen; ++spin) {
            if (spin == spin_max) { atomicExch(a.fault, 1ull); break; }
            __builtin_amdgcn_s_sleep(1); st0 = __hip_atomic_load(poll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t part = (uint32_t)st0;
        if (wv >= 1 && !(a.dbg & 1)) {           // more than BLOCK - 64 groups in front (windows above 2^24 slots): further rounds
            for (int64_t j = tid - 64 + (BLOCK - 64); j < grp; j += BLOCK - 64) {
                unsigned long long st = __hip_atomic_load(gstatus + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned int spin = 0; (st >> 34) != a.gen; ++spin) {
                    if (spin == spin_max) { atomicExch(a.fault, 1ull); break; }
                    __builtin_amdgcn_s_sleep(1); st = __hip_atomic_load(gstatus + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                part += (uint32_t)st;
            }
        }
        part = wave_reduce_add(part);
        if (lane == 0) sSum[wv] = part;
        __syncthreads();      // also: LDS staging complete
        P = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) P += sSum[w];
        if (a.dbg & 5) P = (int64_t)((double)t * (double)a.m / (double)a.ntiles);
    }
    // ---- 4. owned destination offsets (Q0, Q1] ------------------------------------------------------------------------------
    // At 2^20 slots the kernel is bound by instruction issue, not by memory (ablations, round 3: without the stores of this phase
    // 9-10 us of 10.7, without the phase 4.9): offsets are 32-bit, and a group of 128 offsets that the tile owns completely — all
    // but the first and the last one — takes a path without ownership tests, with one gather per occupancy word instead of
    // four bit interleaves.
    const int Q0 = P == 0 ? 0 : dest_of_rank(g, (int)P, a.cells_to_gaps);
    const int Q1 = t == a.ntiles - 1 ? (int)a.Wd : (c == 0 ? Q0 : dest_of_rank(g, (int)(P + c), a.cells_to_gaps));
    if (Q1 <= Q0 || (a.dbg & 2)) return;
    const int Wd = (int)a.Wd, Pi = (int)P;
    key_t* __restrict__ dk = dstk + a.dst_lo0;                       // offset q (1-based) lives in dk[q - 1]
    double* __restrict__ dv = a.dst_vals + a.dst_lo0;
    uint64_t* __restrict__ dw = a.dst_occ + (a.dst_lo0 >> 6);        // (windows of this kernel start on an occupancy word)
    typedef double d2v __attribute__((ext_vector_type(2)));
    typedef key_t k2v __attribute__((ext_vector_type(2)));
    const int g_first = Q0 & ~127, g_last = (Q1 - 1) & ~127;
    for (int gq = g_first + wv * 128; gq <= g_last; gq += NW * 128) {
        const int qa = gq + 2 * lane + 1;                             // 1-based offsets qa, qa + 1
        if (gq >= Q0 && gq + 128 <= Q1) {
            // ---- every offset of the group is owned (and inside the window)
            int k; bool gp0, gp1;
            gap_pair(g, qa, &k, &gp0, &gp1);      // (a straight-line form with one predicated step each way was slower: 9.0 vs 8.8 us at 2^20)
            const int r0 = qa - k, r1 = qa + 1 - k;                   // 1-based ranks of the cells on qa / qa + 1 (when not gaps)
            key_t k0 = 0, k1 = 0;
            double v0 = 0.0, v1 = 0.0;
            if (!gp0) { k0 = sK[r0 - Pi - 1]; v0 = sV[r0 - Pi - 1]; }
            if (!gp1) { k1 = sK[r1 - Pi - 1]; v1 = sV[r1 - Pi - 1]; }
            if (a.sems != nullptr) {                                  // spread! with semaphores  src/moves.jl:160-166
                if (!gp0 && k0 == SEM_KEY) a.sems[(int64_t)v0 - 1] = a.dst_lo0 + qa;
                if (!gp1 && k1 == SEM_KEY) a.sems[(int64_t)v1 - 1] = a.dst_lo0 + qa + 1;
            }
            k2v kv; kv.x = k0; kv.y = k1;
            d2v vv2; vv2.x = v0; vv2.y = v1;
            __builtin_nontemporal_store(kv, reinterpret_cast<k2v*>(dk + qa - 1));
            __builtin_nontemporal_store(vv2, reinterpret_cast<d2v*>(dv + qa - 1));
            // occupancy words: bit i of a word = lane i >> 1 of the half, even offset for even i
            const int occ01 = (gp0 ? 0 : 1) | (gp1 ? 0 : 2);
            const int lo = __shfl(occ01, lane >> 1, 64), hi = __shfl(occ01, 32 + (lane >> 1), 64);
            const uint64_t w0 = __ballot((lo >> (lane & 1)) & 1), w1 = __ballot((hi >> (lane & 1)) & 1);
            if (lane == 0) { dw[gq >> 6] = w0; dw[(gq >> 6) + 1] = w1; }
            continue;
        }
        // ---- the first / the last group of the tile: offsets of the neighbours, the end of the window
        key_t k2[2] = {0, 0};
        double v2[2] = {0.0, 0.0};
        bool o2[2] = {false, false};
        if (qa <= Wd) {
            int k; bool gp[2];
            gap_pair(g, qa, &k, &gp[0], &gp[1]);
            bool own[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const bool gap = gp[jj];
                if (jj == 1 && gap) ++k;
                own[jj] = qa + jj > Q0 && qa + jj <= Q1;
                o2[jj] = !gap && qa + jj <= Wd;
                if (o2[jj] && own[jj]) {
                    const int rank = qa + jj - k;
                    k2[jj] = sK[rank - Pi - 1]; v2[jj] = sV[rank - Pi - 1];
                    if (a.sems != nullptr && k2[jj] == SEM_KEY) a.sems[(int64_t)v2[jj] - 1] = a.dst_lo0 + qa + jj;   // 1-based slot
                }
            }
            if (own[0] && own[1]) {
                k2v kv; kv.x = k2[0]; kv.y = k2[1];
                d2v vv2; vv2.x = v2[0]; vv2.y = v2[1];
                __builtin_nontemporal_store(kv, reinterpret_cast<k2v*>(dk + qa - 1));
                __builtin_nontemporal_store(vv2, reinterpret_cast<d2v*>(dv + qa - 1));
            } else if (own[0]) {
                dk[qa - 1] = k2[0]; dv[qa - 1] = v2[0];
            } else if (own[1]) {
                dk[qa] = k2[1]; dv[qa] = v2[1];
            }
        }
        // occupancy words are closed-form: written by the owner of their first offset
        const uint64_t be = __ballot(o2[0]);
        const uint64_t bo = __ballot(o2[1]);
        if (lane == 0) {
            if (gq + 1 > Q0 && gq + 1 <= Q1)
                dw[gq >> 6] = spread_bits32((uint32_t)be) | (spread_bits32((uint32_t)bo) << 1);
            if (gq + 64 < Wd && gq + 65 > Q0 && gq + 65 <= Q1)
                dw[(gq >> 6) + 1] = spread_bits32((uint32_t)(be >> 32)) | (spread_bits32((uint32_t)(bo >> 32)) << 1);
        }
    }
}

// K-pack as a stand-alone kernel: the occupied cells of a slot range, in slot order, to a dense output (pack! of
// src/moves.jl:94-110 into a separate buffer).  Serves per-column iteration (DynamicMatrixColView, src/views.jl:15-35)
// and iteration over a vector (src/pma.jl:165-180): one wave per 4096-slot source tile, lane <-> occupancy word for the
// prefix, then lane <-> slot with ballot-style popcount ranks; out[tile_off + rank] is a contiguous run per wave.
__global__ __launch_bounds__(64) void k_compact(KeyArr keys, const double* __restrict__ vals,
                                                const uint64_t* __restrict__ occ, int64_t lo0, int64_t hi0, int64_t w0,
                                                int64_t nwords, const uint32_t* __restrict__ tile_off,
                                                KeyArr out_keys, double* __restrict__ out_vals) {
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    const int64_t wl = t * SRC_TILE_WORDS + lane;
    uint64_t myword = 0;
    if (wl < nwords) myword = occ[w0 + wl] & range_mask_for_word(w0 + wl, lo0, hi0);
    const uint32_t myoff = wave_excl_scan((uint32_t)popc64(myword));
    const int64_t base = tile_off[t];
    for (int w = 0; w < SRC_TILE_WORDS; ++w) {
        const uint64_t mask = __shfl(myword, w, 64);
        const int64_t woff = base + (int64_t)__shfl(myoff, w, 64);  // cross-lane reads stay outside divergent code
        if (mask == 0) continue;                                   // wave-uniform
        if ((mask >> lane) & 1ull) {
            const int64_t r = woff + popc64(mask & mask_lt(lane));
            const int64_t s = ((w0 + t * SRC_TILE_WORDS + w) << 6) + lane;
            out_keys[r] = keys[s];
            out_vals[r] = vals[s];
        }
    }
}

hipError_t launch_compact_range(KeyArr keys, const double* vals, const uint64_t* occ, int64_t from, int64_t to,
                                KeyArr out_keys, double* out_vals, int64_t out_cap, RebalanceWork* work, int64_t* count,
                                hipStream_t stream) {
    *count = 0;
    if (to < from) return hipSuccess;
    const int64_t lo0 = from - 1, hi0 = to - 1, w0 = lo0 >> 6;
    const int64_t nwords = (hi0 >> 6) - w0 + 1;
    const int64_t ntiles = (nwords + SRC_TILE_WORDS - 1) / SRC_TILE_WORDS;
    if (ntiles + 1 > work->tiles_cap) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tile_count, dim3((unsigned)((ntiles + CNT_TILES - 1) / CNT_TILES)), dim3(256), 0, stream, occ, lo0, hi0, w0, nwords, ntiles, work->tile_cnt);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, stream, work->tile_cnt, work->tile_off, ntiles);
    uint32_t total = 0;
    hipError_t e = hipMemcpyAsync(&total, work->tile_off + ntiles, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    *count = total;
    if (total == 0) return hipSuccess;
    if ((int64_t)total > out_cap) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_compact, dim3((unsigned)ntiles), dim3(64), 0, stream, keys, vals, occ, lo0, hi0, w0, nwords, work->tile_off,
                       out_keys, out_vals);
    return hipGetLastError();
}

// ---- whole-vector operations on two packed (key, value) streams (the output of K-pack): == and a*x + b*y ----
// v1 == v2 on the stored entries  (src/vector.jl:85-87 -> src/pma.jl:262-266 -> _arrays_equal src/pma.jl:236-260): the i-th
// stored tuples must compare equal as Julia tuples: Int keys ==, Float64 values == (IEEE: NaN differs from itself, -0.0 == 0.0).
__global__ __launch_bounds__(256) void k_packed_equal(KeyArr ka, const double* __restrict__ va, KeyArr kb,
                                                      const double* __restrict__ vb, int64_t n, int32_t* __restrict__ differ) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool d = false;
    if (i < n) d = (ka[i] != kb[i]) || !(va[i] == vb[i]);
    if (__ballot(d) != 0ull && (threadIdx.x & 63) == 0) atomicOr(differ, 1);
}
hipError_t launch_packed_equal(KeyArr ka, const double* va, KeyArr kb, const double* vb, int64_t n, int32_t* differ, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_packed_equal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ka, va, kb, vb, n, differ);
    return hipGetLastError();
}

// merge of two ascending key streams into one array of na + nb cells, equal keys adjacent (x's cell first): thread i places
// x[i] at i + lower_bound(y, key) and y[j] at j + upper_bound(x, key).  A key stored in both operands keeps ONE cell (x's slot,
// value alpha*x + beta*y, dropped when that is zero — the both-stored rule of the sparse-vector +/- the reference falls back
// to, test/functional/math.jl:53-94); one-sided cells carry alpha*x or beta*y.  `keep` is the occupancy bitmap K-pack consumes.
static __device__ __forceinline__ int64_t packed_bound(KeyArr k, int64_t n, int64_t key, bool upper) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t c = k[mid];
        if (c < key || (upper && c == key)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void k_merge_axpby(KeyArr ka, const double* __restrict__ va, int64_t na, double alpha,
                                                     KeyArr kb, const double* __restrict__ vb, int64_t nb, double beta,
                                                     int64_t* __restrict__ mk, double* __restrict__ mv, uint64_t* __restrict__ keep) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < na) {
        const int64_t key = ka[t];
        const int64_t lb = packed_bound(kb, nb, key, false);
        const bool both = lb < nb && kb[lb] == key;
        const double v = both ? __dadd_rn(__dmul_rn(alpha, va[t]), __dmul_rn(beta, vb[lb])) : __dmul_rn(alpha, va[t]);
        const int64_t p = t + lb;
        mk[p] = key; mv[p] = v;
        if (!both || v != 0.0) atomicOr(reinterpret_cast<unsigned long long*>(keep + (p >> 6)), 1ull << (p & 63));
    } else if (t < na + nb) {
        const int64_t j = t - na;
        const int64_t key = kb[j];
        const int64_t ub = packed_bound(ka, na, key, true);
        const bool both = ub > 0 && ka[ub - 1] == key;
        const int64_t p = j + ub;
        mk[p] = key; mv[p] = __dmul_rn(beta, vb[j]);
        if (!both) atomicOr(reinterpret_cast<unsigned long long*>(keep + (p >> 6)), 1ull << (p & 63));
    }
}
hipError_t launch_merge_axpby(KeyArr ka, const double* va, int64_t na, double alpha, KeyArr kb, const double* vb, int64_t nb,
                              double beta, int64_t* mk, double* mv, uint64_t* keep, hipStream_t stream) {
    if (na + nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_axpby, dim3((unsigned)((na + nb + 255) / 256)), dim3(256), 0, stream, ka, va, na, alpha, kb, vb, nb, beta,
                       mk, mv, keep);
    return hipGetLastError();
}

// ---- K-permute: order-preserving move between two occupancy bitmaps ---------------------------------------------------
// The cells of a PMA always keep their relative order, so the layout after ANY sequence of inserts is fully determined by
// the final occupancy bitmap: the r-th cell (old bitmap order, followed by the appended cells in op order) sits at the r-th
// set bit of the new bitmap.  The append fast path of the sequencer simulates pack! / spread! on the BITMAP only (no cell
// moves) and this kernel then moves every cell exactly once.  One workgroup per 4096-slot destination tile; ranks <= n0
// come from the source array (compaction through LDS as in k_move), ranks > n0 from the batch's op records.
constexpr int PERM_TILE = 4096;
struct PermArgs {
    KeyArr src_keys; const double* src_vals; const uint64_t* src_occ; int64_t src_words;
    KeyArr dst_keys; double* dst_vals; const uint64_t* dst_occ; int64_t dst_words;
    const uint32_t* src_off; int64_t src_tiles;        // exclusive prefix per 4096-slot source tile (k_tile_scan)
    const uint32_t* dst_off; int64_t dst_tiles;
    int64_t n0;                                         // cells that existed before the run
    const Op* ops; int64_t i0;                          // appended cell r (1-based beyond n0) = ops[i0 + r - 1]
    int64_t* sems;
};
__global__ __launch_bounds__(256) void k_permute(PermArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char perm_lds[];
    int64_t* sK = reinterpret_cast<int64_t*>(perm_lds);
    double* sV = reinterpret_cast<double*>(perm_lds + PERM_TILE * sizeof(int64_t));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t d = blockIdx.x;
    const int64_t R0 = a.dst_off[d];
    const int64_t cnt = (int64_t)a.dst_off[d + 1] - R0;            // cells landing in this tile: ranks R0+1 .. R0+cnt
    if (cnt == 0) return;
    // ---- stage: ranks from the source array
    const int64_t from_src = R0 < a.n0 ? ((R0 + cnt <= a.n0 ? cnt : a.n0 - R0)) : 0;
    if (from_src > 0) {
        int64_t lo = 0, hi = a.src_tiles - 1;                      // largest t with src_off[t] <= R0
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if ((int64_t)a.src_off[mid] <= R0) lo = mid; else hi = mid - 1;
        }
        for (int64_t t = lo; t < a.src_tiles && (int64_t)a.src_off[t] < R0 + from_src; ++t) {
            const int64_t base = a.src_off[t];
            const int64_t wl = t * SRC_TILE_WORDS + lane;
            const uint64_t myword = wl < a.src_words ? a.src_occ[wl] : 0ull;
            const uint32_t myoff = wave_excl_scan((uint32_t)popc64(myword));
            for (int w = wv; w < SRC_TILE_WORDS; w += 4) {
                const uint64_t mask = __shfl(myword, w, 64);
                const int64_t off = base + (int64_t)__shfl(myoff, w, 64);
                if ((mask >> lane) & 1ull) {
                    const int64_t rank = off + popc64(mask & mask_lt(lane)) + 1;
                    if (rank > R0 && rank <= R0 + from_src) {
                        const int64_t s = ((t * SRC_TILE_WORDS + w) << 6) + lane;
                        sK[rank - R0 - 1] = a.src_keys[s];
                        sV[rank - R0 - 1] = a.src_vals[s];
                    }
                }
            }
        }
    }
    // ---- stage: appended cells
    for (int64_t r = from_src + tid; r < cnt; r += 256) {
        const Op op = a.ops[a.i0 + (R0 + r - a.n0)];
        sK[r] = op.a; sV[r] = op.v;
    }
    __syncthreads();
    // ---- write: lane <-> slot, one destination word per wave iteration
    const int64_t wbase = d * SRC_TILE_WORDS;
    const int64_t wl = wbase + lane;
    const uint64_t dword = wl < a.dst_words ? a.dst_occ[wl] : 0ull;
    const uint32_t doff = wave_excl_scan((uint32_t)popc64(dword));
    for (int w = wv; w < SRC_TILE_WORDS; w += 4) {
        const uint64_t mask = __shfl(dword, w, 64);
        const uint32_t off = __shfl(doff, w, 64);
        if ((mask >> lane) & 1ull) {
            const uint32_t r = off + (uint32_t)popc64(mask & mask_lt(lane));
            const int64_t pos0 = ((wbase + w) << 6) + lane;
            const int64_t k = sK[r];
            const double v = sV[r];
            a.dst_keys[pos0] = k;
            a.dst_vals[pos0] = v;
            if (a.sems != nullptr && k == SEM_KEY) a.sems[(int64_t)v - 1] = pos0 + 1;
        }
    }
}

hipError_t launch_permute(KeyArr src_keys, const double* src_vals, const uint64_t* src_occ, int64_t src_cap,
                          KeyArr dst_keys, double* dst_vals, const uint64_t* dst_occ, int64_t dst_cap, int64_t n0,
                          const Op* ops, int64_t i0, int64_t* sems, RebalanceWork* wsrc, RebalanceWork* wdst, hipStream_t stream) {
    PermArgs a;
    a.src_keys = src_keys; a.src_vals = src_vals; a.src_occ = src_occ; a.src_words = (src_cap + 63) / 64;
    a.dst_keys = dst_keys; a.dst_vals = dst_vals; a.dst_occ = dst_occ; a.dst_words = (dst_cap + 63) / 64;
    a.src_tiles = (a.src_words + SRC_TILE_WORDS - 1) / SRC_TILE_WORDS;
    a.dst_tiles = (a.dst_words + SRC_TILE_WORDS - 1) / SRC_TILE_WORDS;
    if (a.src_tiles + 1 > wsrc->tiles_cap || a.dst_tiles + 1 > wdst->tiles_cap) return hipErrorInvalidValue;
    // tile prefixes of both bitmaps: one count launch and one scan launch for the two of them
    const int nbA = (int)((a.src_tiles + CNT_TILES - 1) / CNT_TILES), nbB = (int)((a.dst_tiles + CNT_TILES - 1) / CNT_TILES);
    hipLaunchKernelGGL(k_tile_count2, dim3((unsigned)(nbA + nbB)), dim3(256), 0, stream, src_occ, src_cap - 1, a.src_words, a.src_tiles, wsrc->tile_cnt, nbA,
                       dst_occ, dst_cap - 1, a.dst_words, a.dst_tiles, wdst->tile_cnt);
    hipLaunchKernelGGL(k_tile_scan, dim3(2), dim3(1024), 0, stream, (const uint32_t*)wsrc->tile_cnt, wsrc->tile_off, a.src_tiles,
                       (const uint32_t*)wdst->tile_cnt, wdst->tile_off, a.dst_tiles);
    a.src_off = wsrc->tile_off; a.dst_off = wdst->tile_off;
    a.n0 = n0; a.ops = ops; a.i0 = i0; a.sems = sems;
    const size_t lds = (size_t)PERM_TILE * 16;
    static PerDeviceOnce once;
    {
        hipError_t e = once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(k_permute), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_permute, dim3((unsigned)a.dst_tiles), dim3(256), lds, stream, a);
    return hipGetLastError();
}

// int32 -> int64 key array (the structure receives its first key outside Int32)
__global__ void k_widen_keys(const int32_t* __restrict__ src, int64_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (int64_t)src[i];
}
hipError_t launch_widen_keys(const void* src32, void* dst64, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_widen_keys, dim3(blocks), dim3(256), 0, stream, (const int32_t*)src32, (int64_t*)dst64, n);
    return hipGetLastError();
}

struct FreshInit { uint64_t* occ0; uint64_t* occ1; int64_t occ_words; unsigned long long* status; int64_t status_words; Ctl* d_ctl; Ctl ctl; };
__global__ __launch_bounds__(256) void k_init_fresh(FreshInit a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.occ_words; i += stride) { a.occ0[i] = 0ull; a.occ1[i] = 0ull; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.status_words; i += stride) a.status[i] = 0ull;
    if (blockIdx.x == 0) {
        static_assert(sizeof(Ctl) % 8 == 0, "Ctl is copied in 8-byte words");
        const int64_t* src = reinterpret_cast<const int64_t*>(&a.ctl);
        int64_t* dst = reinterpret_cast<int64_t*>(a.d_ctl);
        for (int i = threadIdx.x; i < (int)(sizeof(Ctl) / 8); i += blockDim.x) dst[i] = src[i];
    }
}
// the control block written from a launch argument (stream-ordered, no copy command: the pinned mirror is not a DMA source)
struct CtlStore { Ctl* d_ctl; Ctl ctl; };
__global__ __launch_bounds__(256) void k_store_ctl(CtlStore a) {
    const int64_t* src = reinterpret_cast<const int64_t*>(&a.ctl);
    int64_t* dst = reinterpret_cast<int64_t*>(a.d_ctl);
    for (int i = threadIdx.x; i < (int)(sizeof(Ctl) / 8); i += blockDim.x) dst[i] = src[i];
}
hipError_t launch_store_ctl(Ctl* d_ctl, const Ctl& ctl, hipStream_t stream) {
    CtlStore a{d_ctl, ctl};
    hipLaunchKernelGGL(k_store_ctl, dim3(1), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_init_fresh(uint64_t* occ0, uint64_t* occ1, int64_t occ_words, unsigned long long* status, int64_t status_words,
                             Ctl* d_ctl, const Ctl& ctl, hipStream_t stream) {
    FreshInit a{occ0, occ1, occ_words, status, status_words, d_ctl, ctl};
    const int64_t n = std::max(occ_words, status_words);
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), 1024);
    hipLaunchKernelGGL(k_init_fresh, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

__global__ void k_clear_occ(uint64_t* occ, int64_t lo0, int64_t hi0) {
    const int64_t w0 = lo0 >> 6, w1 = hi0 >> 6;
    for (int64_t w = w0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w <= w1; w += (int64_t)gridDim.x * blockDim.x)
        occ[w] &= ~range_mask_for_word(w, lo0, hi0);
}

hipError_t launch_clear_occ(uint64_t* occ, int64_t from, int64_t to, hipStream_t stream) {
    if (to < from) return hipSuccess;
    const int64_t nwords = ((to - 1) >> 6) - ((from - 1) >> 6) + 1;
    const int blocks = (int)((nwords + 255) / 256 > 1024 ? 1024 : (nwords + 255) / 256);
    hipLaunchKernelGGL(k_clear_occ, dim3(blocks), dim3(256), 0, stream, occ, from - 1, to - 1);
    return hipGetLastError();
}

// bench hook (dsa_vec_dev_relayout mode 2): cells packed to the right end of the array — "all gaps at the left"
__global__ __launch_bounds__(256) void k_pack_right(KeyArr src_keys, const double* __restrict__ src_vals, int64_t m, KeyArr dst_keys,
                                                    double* __restrict__ dst_vals, uint64_t* __restrict__ dst_occ, int64_t cap) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;          // destination slot (0-based); cap is a multiple of 64 or < 64
    const bool in = s < cap && s >= cap - m;
    if (in) { dst_keys[s] = (int64_t)src_keys[s - (cap - m)]; dst_vals[s] = src_vals[s - (cap - m)]; }
    const uint64_t b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && s < cap) dst_occ[s >> 6] = b;
}
hipError_t launch_pack_right(KeyArr src_keys, const double* src_vals, int64_t m, KeyArr dst_keys, double* dst_vals, uint64_t* dst_occ,
                             int64_t cap, hipStream_t stream) {
    hipLaunchKernelGGL(k_pack_right, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream, src_keys, src_vals, m, dst_keys, dst_vals, dst_occ, cap);
    return hipGetLastError();
}

hipError_t launch_rebalance(KeyArr src_keys, const double* src_vals, const uint64_t* src_occ,
                            int64_t src_ws, int64_t src_we, bool src_packed,
                            KeyArr dst_keys, double* dst_vals, uint64_t* dst_occ,
                            int64_t dst_ws, int64_t dst_we, int64_t m, int64_t* sems,
                            RebalanceWork* work, hipStream_t stream) {
    {
        Move2Args a;
        a.src_keys = src_keys; a.src_vals = src_vals; a.src_occ = src_occ;
        a.src_lo0 = src_ws - 1; a.src_hi0 = src_we - 1;
        a.dst_keys = dst_keys; a.dst_vals = dst_vals; a.dst_occ = dst_occ;
        a.dst_lo0 = dst_ws - 1;
        a.Wd = dst_we - dst_ws + 1; a.m = m;
        a.sems = sems;
        a.cells_to_gaps = m > 0 ? (double)(a.Wd - m) / (double)m : 0.0;
        { const SpreadGeom hg = make_geom(a.Wd, m); a.geom_f = hg.f; a.geom_inv_f = hg.inv_f; }
        { static const char* e = dev_env("DSA_DBG_MOVE2"); a.dbg = e ? atoi(e) : 0; }
        if (((src_ws - 1) & 63) != 0 && !src_packed) return hipErrorInvalidValue;     // source windows start on an occupancy word
        // ranks, gap indices and offsets inside a window are 32-bit in the kernel (dest_of_rank, gap_pair: single-instruction
        // int <-> double conversions): windows of 2^31 slots or more are refused, not wrapped
        if (a.Wd >= ((int64_t)1 << 31) || src_we - src_ws + 1 >= ((int64_t)1 << 31) || m >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
        if (src_packed) {
            a.ntiles = std::max<int64_t>(1, (m + M2_TILE_BIG - 1) / M2_TILE_BIG);
            a.status = nullptr; a.gen = 0; a.fault = nullptr;
            // (512 threads up to 8192 tiles like the general source)
            static const int force_block_p = [] { const char* e = dev_env("DSA_MOVE2_BLOCK"); return e ? atoi(e) : 0; }();
            const int block = force_block_p ? force_block_p : (a.ntiles <= 8192 ? 512 : 256);
            if (block == 512) {
                if (a.src_keys.wide) hipLaunchKernelGGL((k_move2<true, true, 512, M2_TILE_BIG>), dim3((unsigned)a.ntiles), dim3(512), 0, stream, a);
                else hipLaunchKernelGGL((k_move2<true, false, 512, M2_TILE_BIG>), dim3((unsigned)a.ntiles), dim3(512), 0, stream, a);
            } else {
                if (a.src_keys.wide) hipLaunchKernelGGL((k_move2<true, true, 256, M2_TILE_BIG>), dim3((unsigned)a.ntiles), dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((k_move2<true, false, 256, M2_TILE_BIG>), dim3((unsigned)a.ntiles), dim3(256), 0, stream, a);
            }
            return hipGetLastError();
        }
        const int64_t Ws = src_we - src_ws + 1;
        // up to 2^24 slots: 8 waves per tile; above: 4.
        // Tiles of 1024 slots (twice the workgroups, half the write phase each; DSA_MOVE2_TILE=1024) were measured in round 3 and LOSE:
        // 2^20 slots 11.1 vs 10.1 us, 2^21 18.0 vs 15.0, 2^22 26.1 vs 25.2, 2^24 76.7 vs 77.7 — more status words to publish and poll.  Tiles of
        // 4096 slots (half the per-tile fixed work) bought nothing either: 8.6 vs 8.2 us at 2^20 with 512 threads, 8.0 with 1024 (and 72.6 vs 69 at 2^24)
        static const int force_block = [] { const char* e = dev_env("DSA_MOVE2_BLOCK"); return e ? atoi(e) : 0; }();
        static const int force_tile = [] { const char* e = dev_env("DSA_MOVE2_TILE"); return e ? atoi(e) : 0; }();
        const int tile = force_tile == M2_TILE_SMALL ? M2_TILE_SMALL : M2_TILE_BIG;
        a.ntiles = (Ws + tile - 1) / tile;
        if (a.ntiles + (a.ntiles >> 6) + 2 > work->status_cap || work->status == nullptr) return hipErrorInvalidValue;
        if (++work->gen >= (1ull << 30)) {          // generation wrap: start over on a zeroed table
            hipError_t e = hipMemsetAsync(work->status, 0, (size_t)work->status_cap * sizeof(unsigned long long), stream);
            if (e != hipSuccess) return e;
            work->gen = 1;
        }
        a.status = work->status; a.gen = work->gen;
        a.fault = work->status + work->status_cap - 1;              // the last word of the table is not a status word (alloc_work)
        // measured, 512 vs 256 threads (after the write phase lost two thirds of its instructions): 2^20 8.8 vs 10.4 us, 2^22 22.7 vs 25.0,
        // 2^23 38.5 vs 41.4, 2^24 69.2-70.8 vs 72.6-75.3, 2^25 135-141 vs 136-139, 2^26 285-296 vs 270-281 (1024 threads lose everywhere)
        const int block = force_block ? force_block : (a.ntiles <= 8192 ? 512 : 256);
#define DSA_MOVE2_LAUNCH(W_, B_, T_) hipLaunchKernelGGL((k_move2<false, W_, B_, T_>), dim3((unsigned)a.ntiles), dim3(B_), 0, stream, a)
        const bool wide = a.src_keys.wide != 0;
        if (tile == M2_TILE_SMALL) {
            if (block == 512) { if (wide) DSA_MOVE2_LAUNCH(true, 512, M2_TILE_SMALL); else DSA_MOVE2_LAUNCH(false, 512, M2_TILE_SMALL); }
            else { if (wide) DSA_MOVE2_LAUNCH(true, 256, M2_TILE_SMALL); else DSA_MOVE2_LAUNCH(false, 256, M2_TILE_SMALL); }
        } else {
            if (block == 512) { if (wide) DSA_MOVE2_LAUNCH(true, 512, M2_TILE_BIG); else DSA_MOVE2_LAUNCH(false, 512, M2_TILE_BIG); }
            else { if (wide) DSA_MOVE2_LAUNCH(true, 256, M2_TILE_BIG); else DSA_MOVE2_LAUNCH(false, 256, M2_TILE_BIG); }
        }
#undef DSA_MOVE2_LAUNCH
        return hipGetLastError();
    }
}

}  // namespace dsa
