// csrc/spmv.hip — K-spmv: PCSR sparse matrix x dense vector on gfx950.
//
// Reproduces _mul / _mul_dyn_mat_col_loop! (src/operations.jl:62-135) behind the `*` methods
// (src/operations.jl:14-60) for a dense x (every index stored).  Two forms over ONE orientation
// P of the matrix (a MappedPackedCSC: slot array + semaphores + partition keys):
//
//   gather  : y[part_key(p)] = sum_{slots of partition p} val * x[key]      (k_spmv_gather; no atomics inside a row)
//             used on the TWIN orientation: mat*v walks rowmajor, transpose(mat)*v walks colmajor.
//             Per output row the terms are added left to right in ascending key order — exactly
//             the reference's accumulation order — except for rows longer than a wave's span
//             (partial sums joined by fp64 atomics; within the 1e-12 tolerance).
//   scatter : y[key] += x[part_key(p)] * val with fp64 atomics — the literal loop nest of the
//             reference on its own orientation (mat*v walks colmajor)                       (k_spmv_scatter).
//
// Bound: HBM / fabric.  Algorithmic bytes per launch = 16*capacity + 8*nx + 8*ny: the flat
// scan streams every slot (gaps included — they are part of the bit-identical layout) once.
#include "dsa_dev.h"
#include <type_traits>

namespace dsa {

constexpr int SP_TILE = 2048;
constexpr int SP_BLOCK = 256;
constexpr int SP_WAVES = SP_BLOCK / 64;
constexpr int SP_PER_WAVE = SP_TILE / SP_WAVES;     // 512 slots = 8 words
constexpr int SP_WORDS_PER_WAVE = SP_PER_WAVE / 64;
constexpr int SP_BACK_WORDS = 32;                    // backward ballot scan limit before the table bisection

__device__ __forceinline__ double wave_reduce_add_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// partition id (1-based) whose semaphore is the last one located at a 0-based slot < b0, or 0.
// Executed by one full wave.
__device__ int64_t carry_in_partition(KeyArr keys, const double* __restrict__ vals,
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

// scatter form: one 256-thread workgroup per 2048-slot tile; wave w owns 8 occupancy words, lane <-> slot; values and
// semaphore ids are staged in LDS, every cell finds the semaphore in front of it from the per-word semaphore ballots.
__global__ __launch_bounds__(SP_BLOCK) void k_spmv_scatter(KeyArr keys, const double* __restrict__ vals,
                                                   const uint64_t* __restrict__ occ, int64_t capacity,
                                                   const int64_t* __restrict__ sems,
                                                   const int64_t* __restrict__ part_keys, int64_t table_len,
                                                   const double* __restrict__ x, int64_t nx,
                                                   double* __restrict__ y, int64_t ny, int pattern) {
    __shared__ double sP[SP_TILE];
    __shared__ uint16_t sSemList[SP_WAVES][SP_PER_WAVE];
    __shared__ int sSemCnt[SP_WAVES];
    __shared__ int64_t sCarry;
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

    // ---- phase 1: stream the tile, values -> LDS ---------------------------------------------------------
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
            if (bit[j]) { k[j] = keys.ld_nt(s); v[j] = __builtin_nontemporal_load(vals + s); }
        }
    }
    int nsem = 0;
#pragma unroll
    for (int j = 0; j < SP_WORDS_PER_WAVE; ++j) {
        const int ls = wv * SP_PER_WAVE + j * 64 + lane;
        const bool issem = bit[j] && k[j] == SEM_KEY;
        sP[ls] = bit[j] ? v[j] : 0.0;
        const uint64_t sb = __ballot(issem);
        if (issem) sSemList[wv][nsem + popc64(sb & mask_lt(lane))] = (uint16_t)ls;
        nsem += popc64(sb);
        if (lane == 0) sSemBits[wv * SP_WORDS_PER_WAVE + j] = sb;
    }
    if (lane == 0) sSemCnt[wv] = nsem;
    __syncthreads();

    int cnt[SP_WAVES];
#pragma unroll
    for (int w = 0; w < SP_WAVES; ++w) cnt[w] = sSemCnt[w];
    int first_sem = tile_end;
#pragma unroll
    for (int w = SP_WAVES - 1; w >= 0; --w) if (cnt[w] > 0) first_sem = sSemList[w][0];

    // the partition that owns the slots in front of the first semaphore of the tile
    if (wv == 0 && (first_sem > 0)) {
        const int64_t c = carry_in_partition(keys, vals, occ, sems, table_len, b0);
        if (lane == 0) sCarry = c;
    }

    {
        // ---- phase 2: every cell finds the semaphore that precedes it -------------------------------
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

// ---- gather form, wave-level ---------------------------------------------------------------------------------------
// One wave per 512 contiguous slots (8 occupancy words); no workgroup barrier.  The cost of this kernel is the number of
// cache-line requests a CU can keep in flight (tools/gatherbench.hip: the slot streams and the x gathers queue for the
// same miss slots of the CU and their times add), so everything a wave will need is requested up front in ONE round of
// straight-line code: 9 words of (key, value) — its own 8 plus the word behind them —, the keys of the word in front,
// then one gather per lane and word: x[key] for a cell (the row key of a partition is fetched once per row by the lane
// that sums it).  The products go to the wave's own LDS slice; the semaphores of the span are compacted (ballot +
// popcount) and one lane per semaphore sums its row from LDS left to right — the reference's accumulation order (src/operations.jl:101), so a row comes
// out bit-identical to the reference unless it is longer than a span.  Who writes a row is decided by where its semaphore is:
//   * a wave writes every row whose semaphore lies in its 512 slots, running past its end (at most one more span,
//     SW_WORDS words) until the next semaphore: plain store, each y entry written once;
//   * the cells in front of a wave's first semaphore belong to an earlier wave — covered by that wave's overrun iff the
//     previous span contains a semaphore; otherwise (rows longer than a span) the wave adds its share with an fp64
//     atomic to the row found by the backward search, and the owner, whose overrun hit the limit, adds with an atomic too.
constexpr int SW_WORDS = 8;
constexpr int SW_SLOTS = (SW_WORDS + 1) * 64;     // products kept per wave: own span + the word behind it
constexpr int SW_COOP = 4;                        // spans with at most this many semaphores: a whole wave per row

__device__ __forceinline__ double product_of(double v, double xv, bool count_pass) {
    // count pass (touched rows of _mul, src/operations.jl:101): 1 for every cell whose x entry is stored (non-zero)
    return count_pass ? (xv != 0.0 ? 1.0 : 0.0) : v * xv;
}

// NT: the slot streams are loaded non-temporal (x larger than an XCD's L2: the stream must not evict it) or plain (x
//     L2-resident: plain stream loads overlap the gathers better; tools/gatherbench2.hip: 56.8 vs 74.5 us at 1 MB of x).
// ZFILL: y is NOT zeroed in front of the launch.  Every row that owns a semaphore is written exactly once (plain
//     store) and the owner of a semaphore also zeroes the rows between the previous partition key and its own (the
//     last partition: up to ny).  Only valid when no partition is longer than a span (no atomics), the tables are in
//     key order without tombstones, and at least one partition exists — decided by the host (SpmvMeta).
// ZFILL: rows without a partition between the previous partition key and `row` (exclusive); behind the last partition: up to ny
__device__ __forceinline__ void zero_fill_front(double* __restrict__ y, int64_t ny, int64_t prev, int64_t row, bool tail) {
    const int64_t lo = prev > 0 ? prev : 0;
    const int64_t hi = row < ny + 1 ? row : ny + 1;
    for (int64_t r = lo + 1; r < hi; ++r) y[r - 1] = 0.0;
    if (tail) for (int64_t r = (row > 0 ? row : 0) + 1; r <= ny; ++r) y[r - 1] = 0.0;
}

// SHARE: the four waves of a workgroup hold four CONSECUTIVE spans, so the word behind the span of waves 0..2 is the first word of the
//     next wave, whose products are in LDS anyway: after one workgroup barrier a wave walks on into its neighbour's slice instead of
//     loading (and gathering for) a ninth word of its own, and learns from the neighbour in front whether that span holds a semaphore
//     instead of loading the keys of the word in front.  Only wave 3 loads a ninth word, only wave 0 the word in front: 33 + 1 words
//     of requests per workgroup instead of 36 + 4.  Needs every wave of the grid alive at the barrier: capacity a multiple of SP_TILE
//     (decided by launch_spmv).  Same sums in the same order as without it.
// the work of ONE tile (4 spans, one per wave); `tile` is the tile index after the XCD mapping of the caller
template <bool WIDE, bool NT, bool ZFILL, bool SHARE>
__device__ __forceinline__ void gather_tile(KeyArr keys, const double* __restrict__ vals,
                                            const uint64_t* __restrict__ occ, int64_t capacity,
                                            const int64_t* __restrict__ sems,
                                            const int64_t* __restrict__ part_keys, int64_t table_len,
                                            const double* __restrict__ x, int64_t nx,
                                            double* __restrict__ y, int64_t ny, int pattern, int64_t tile, int tid) {
    typedef typename std::conditional<WIDE, int64_t, int32_t>::type key_t;      // physical key width, fixed at compile time for the streams
    const key_t* __restrict__ kp = static_cast<const key_t*>(keys.p);
    // Products: with SHARE the 4 spans + the word behind the last one, FLAT (wave w at w * 512, running on into wave w + 1's slice);
    // without it every wave keeps its own ninth word.  ~21 KB per workgroup: 7 workgroups per CU (8 were measured with a shorter
    // semaphore list: no difference, the kernel is not occupancy-bound).
    __shared__ double sPw[SHARE ? SP_WAVES * SW_WORDS * 64 + 64 : SP_WAVES * SW_SLOTS];
    __shared__ uint16_t sListw[SP_WAVES][SW_WORDS * 64];
    __shared__ uint64_t sSb0[SP_WAVES];                 // SHARE: semaphore ballot of a wave's first word, number of semaphores of its span
    __shared__ int sNsem[SP_WAVES];
    // Occupancy limiter of the non-temporal instantiations: 16 KB of LDS nobody uses make it 4 workgroups per CU instead of 7.  The kernel
    // does not live on occupancy (2 per CU run config 3 within 2 %): fewer waves queueing in front of the CU's texture addresser were
    // measured FASTER where the gathers hit (banded shape 84-87 vs 90 us, same box, twice) and equal on config 3 (117.6 vs 117.8).
    // 3 per CU: config 3 +2 %; 5 per CU: no gain.
    __shared__ double sPad[NT ? 2048 : 1];
    if (capacity == -2) { sPad[tid] = 1.0; __syncthreads(); y[0] = sPad[0]; }        // (never true: keeps the array allocated)
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // SHARE: one flat product array over the same storage — wave w's slots at [w * SW_WORDS * 64, ...), running on into wave w + 1's; behind
    // the four spans the ninth word of wave 3, then the four front keys
    constexpr int SPAN = SW_WORDS * 64;
    double* sP = SHARE ? sPw + wv * SPAN : sPw + wv * SW_SLOTS;
    const bool ninth = !SHARE || wv == SP_WAVES - 1;    // this wave loads the word behind its span itself
    uint16_t* sList = sListw[wv];
    const int64_t nwords = (capacity + 63) >> 6;      // slot buffers are allocated in whole words (cap_alloc >= 4096)
    const int64_t w0 = tile * (SP_TILE / 64) + (int64_t)wv * SW_WORDS;
    if (w0 >= nwords) return;
    const bool count_pass = (pattern & 3) == 1;

    // ---- one round of requests (straight-line: word indices are clamped, not predicated) ---------------------------------
    // the occupancy words of the span and of the word behind it: uniform loads, all of them issued before anything waits for one
    // (round 4 loaded them one by one in front of each word's gathers: nine dependent scalar round trips per wave)
    uint64_t ow[SW_WORDS + 1];
    key_t k[SW_WORDS + 1];
    double v[SW_WORDS + 1];
    uint64_t pword = 0ull;
    key_t pk = 0;
    if (!NT && w0 > 0 && w0 + SW_WORDS < nwords) {
        // a span inside the array (all but the first and the last one): no clamping, ONE base address per stream and constant offsets —
        // the general form below spends ~100 scalar and vector instructions per wave on index arithmetic.  Only where x is L2-resident
        // (plain streams): the final config-5 product 8.2 -> 7.7 us; on config 3, whose gathers miss, the tighter burst of stream
        // loads is the SLOWER form (118.5 vs 117.8 us, same box, twice)
        const uint64_t* __restrict__ op = occ + w0;
        const key_t* __restrict__ kb = kp + (w0 << 6) + lane;
        const double* __restrict__ vb = vals + (w0 << 6) + lane;
#pragma unroll
        for (int j = 0; j <= SW_WORDS; ++j) ow[j] = op[j];
        // keys first: the gathers wait for them, the values are not needed before the products
#pragma unroll
        for (int j = 0; j <= SW_WORDS; ++j) {
            if (SHARE && j == SW_WORDS && !ninth) { k[j] = 0; continue; }
            k[j] = kb[j * 64];
        }
        if (!SHARE || wv == 0) { pword = op[-1]; pk = kb[-64]; }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j <= SW_WORDS; ++j) {
            if (SHARE && j == SW_WORDS && !ninth) { v[j] = 0.0; continue; }
            v[j] = vb[j * 64];
        }
    } else {
#pragma unroll
        for (int j = 0; j <= SW_WORDS; ++j) ow[j] = occ[w0 + j < nwords ? w0 + j : nwords - 1];
#pragma unroll
        for (int j = 0; j <= SW_WORDS; ++j) {
            if (SHARE && j == SW_WORDS && !ninth) { k[j] = 0; v[j] = 0.0; continue; }
            const int64_t w = w0 + j < nwords ? w0 + j : nwords - 1;
            k[j] = NT ? __builtin_nontemporal_load(kp + (w << 6) + lane) : kp[(w << 6) + lane];
            v[j] = NT ? __builtin_nontemporal_load(vals + (w << 6) + lane) : vals[(w << 6) + lane];
        }
        // the word in front: does the previous span own the cells before our first semaphore?  (SHARE: waves 1..3 ask their neighbour)
        const int64_t pw = w0 > 0 ? w0 - 1 : 0;
        if (!SHARE || wv == 0) {
            pword = w0 > 0 ? occ[pw] : 0ull;
            pk = kp[(pw << 6) + lane];
        }
#pragma unroll
        for (int j = 0; j <= SW_WORDS; ++j) if (w0 + j >= nwords) ow[j] = 0ull;
    }

    // ---- one x gather per lane and word -------------------------------------------------------------------------------------
    // The occupancy bit is folded into the key first (a gap becomes key -1: neither a cell nor a semaphore), so that the occupancy
    // words die here and every later question — cell? semaphore? — is ONE compare on the key whose result is the mask (v_cmp -> SGPR
    // pair), asked again where it is needed instead of being kept: no mask word lives from this loop to the next one (27 of them did
    // until round 5, which is what a tile loop around this body spilled).  Semaphores do not gather here: the row key of a partition is
    // fetched once per ROW by the lane that sums it (below), together with the key of the partition in front of it (ZFILL).
    double xq[SW_WORDS + 1];
    const double* xs = nx > 0 ? x : (const double*)occ;          // what idle lanes read: the first 8 bytes of something that exists
    const uint32_t nx32 = nx < 0x7fffffff ? (uint32_t)(nx > 0 ? nx : 0) : 0x7fffffffu;
    const uint32_t tlen = table_len < 0x7fffffff ? (uint32_t)table_len : 0x7fffffffu;
    auto is_cell = [&](key_t key) -> bool {                      // 1 <= key <= nx (a semaphore and a gap wrap around)
        if (WIDE) return (uint64_t)((int64_t)key - 1) < (uint64_t)(nx > 0 ? nx : 0);
        return (uint32_t)key - 1u < nx32;
    };
#pragma unroll
    for (int j = 0; j <= SW_WORDS; ++j) {
        if (SHARE && j == SW_WORDS && !ninth) { k[j] = (key_t)-1; xq[j] = 0.0; continue; }
        k[j] = __builtin_amdgcn_inverse_ballot_w64(ow[j]) ? k[j] : (key_t)-1;
        if (j == SW_WORDS) {
            // the word behind the span: only the cells in front of its first semaphore matter (and that semaphore itself)
            const uint64_t s8 = __ballot(k[j] == SEM_KEY);
            const uint64_t keep = s8 ? ((s8 & (0ull - s8)) << 1) - 1ull : ~0ull;
            k[j] = __builtin_amdgcn_inverse_ballot_w64(keep) ? k[j] : (key_t)-1;
        }
        const bool cell = is_cell(k[j]);
        if (WIDE) { const int64_t idx = cell ? (int64_t)k[j] - 1 : 0; xq[j] = xs[idx]; }
        else { const uint32_t idx = cell ? (uint32_t)k[j] - 1u : 0u; xq[j] = xs[idx]; }
    }
    // ---- products -> LDS ; the semaphores of the span are compacted, their slot keeps the partition id (the stored Float64) -------
    int nsem = 0;
    uint64_t sb0 = 0ull;
#pragma unroll
    for (int j = 0; j <= SW_WORDS; ++j) {
        if (SHARE && j == SW_WORDS && !ninth) continue;
        const bool cell = is_cell(k[j]);
        const double p = product_of(v[j], xq[j], count_pass);
        sP[j * 64 + lane] = cell ? p : 0.0;
        if (j < SW_WORDS) {
            const uint64_t sbj = __ballot(k[j] == SEM_KEY);
            if (j == 0) sb0 = sbj;
            if (sbj != 0ull) {
                if (__builtin_amdgcn_inverse_ballot_w64(sbj)) {
                    const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sbj >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sbj, 0u));
                    sP[j * 64 + lane] = v[j];            // partition ids are stored as Float64 (src/pcsr.jl:104)
                    sList[nsem + r] = (uint16_t)(j * 64 + lane);
                }
                nsem += popc64(sbj);
            }
        }
    }
    uint64_t sb8 = (SHARE && !ninth) ? 0ull : __ballot(k[SW_WORDS] == SEM_KEY);      // first semaphore behind the span
    const int own_words = (int)(nwords - w0 < SW_WORDS ? nwords - w0 : SW_WORDS);
    // end of the row that is open at the end of the span, as far as the 9 words show it
    const bool behind_valid = w0 + SW_WORDS < nwords;
    if (SHARE) {
        if (lane == 0) { sSb0[wv] = sb0; sNsem[wv] = nsem; }
        __syncthreads();
        if (!ninth) sb8 = sSb0[wv + 1];
    }
    const int endpos = !behind_valid ? own_words * 64 : (sb8 ? SW_WORDS * 64 + __ffsll((unsigned long long)sb8) - 1 : SW_SLOTS);
    bool closed = !behind_valid || sb8 != 0;
    if (!SHARE) __builtin_amdgcn_wave_barrier();

    // row key of the partition whose id sits in slot a of the span (0: no such partition) and — ZFILL — the key of the partition in
    // front of it (ids ascend along the array and the tables are in key order there) and whether it is the last of the table
    struct SemRow { int64_t row, prev; bool last; };
    auto sem_row = [&](int a) -> SemRow {
        const uint32_t id1 = (uint32_t)((int)sP[a] - 1);
        const bool ok = id1 < tlen;
        SemRow r;
        r.row = 0; r.prev = 0; r.last = false;
        if (ok) r.row = part_keys[id1];              // (an id outside the table — there is none in a consistent structure — reads nothing)
        if (ZFILL) { if (ok && id1 > 0) r.prev = part_keys[id1 - 1u]; r.last = ok && id1 + 1u == tlen; }
        return r;
    };

    double open_sum = 0.0;         // last row of the span (wave-uniform after the walk)
    int64_t open_row = 0;
    if (nsem > SW_COOP) {
        // one lane per semaphore, left to right; the row key is requested first and needed last
        for (int e = lane; e < nsem; e += 64) {
            const int a = sList[e];
            const bool is_last = e == nsem - 1;
            const int end = is_last ? endpos : (int)sList[e + 1];
            const SemRow sr = sem_row(a);
            double sum = 0.0;
            int t = a + 1;
            for (; t + 3 < end; t += 4) {
                const double t0 = sP[t], t1 = sP[t + 1], t2 = sP[t + 2], t3 = sP[t + 3];
                sum = sum + t0; sum = sum + t1; sum = sum + t2; sum = sum + t3;
            }
            for (; t < end; ++t) sum = sum + sP[t];
            if (ZFILL) zero_fill_front(y, ny, sr.prev, sr.row, is_last && sr.last);
            if (!is_last || closed) { if (sr.row >= 1 && sr.row <= ny) y[sr.row - 1] = sum; }
        }
        if (!closed) {        // re-sum the open row cooperatively for the slow path below
            const int a = sList[nsem - 1];
            open_row = sem_row(a).row;
            double sum = 0.0;
            for (int t = a + 1 + lane; t < endpos; t += 64) sum += sP[t];
            open_sum = wave_reduce_add_f64(sum);
        }
    } else {
        // few, long rows: the whole wave sums each of them
        for (int e = 0; e < nsem; ++e) {
            const int a = sList[e];
            const bool is_last = e == nsem - 1;
            const int end = is_last ? endpos : (int)sList[e + 1];
            const SemRow sr = sem_row(a);
            if (ZFILL && lane == 0) zero_fill_front(y, ny, sr.prev, sr.row, is_last && sr.last);
            double sum = 0.0;
            for (int t = a + 1 + lane; t < end; t += 64) sum += sP[t];
            sum = wave_reduce_add_f64(sum);
            if (!is_last || closed) { if (lane == 0 && sr.row >= 1 && sr.row <= ny) y[sr.row - 1] = sum; }
            else { open_row = sr.row; open_sum = sum; }
        }
    }

    if (nsem > 0 && !closed) {
        // ---- rows longer than a word behind the span: keep going, at most to the end of the next span ---------------
        for (int64_t w = w0 + SW_WORDS + 1; !closed && w < w0 + 2 * SW_WORDS; ++w) {
            if (w >= nwords) { closed = true; break; }
            const uint64_t wd = occ[w];
            const bool bit = (wd >> lane) & 1ull;
            int64_t kk = -1; double vv = 0.0;
            if (bit) { kk = keys[(w << 6) + lane]; vv = vals[(w << 6) + lane]; }
            const uint64_t sbw = __ballot(bit && kk == SEM_KEY);
            const int lim = sbw ? __ffsll((unsigned long long)sbw) - 1 : 64;
            double pp = 0.0;
            if (bit && lane < lim && kk >= 1 && kk <= nx) pp = product_of(vv, x[kk - 1], count_pass);
            open_sum += wave_reduce_add_f64(pp);
            closed = sbw != 0;
        }
        if (w0 + 2 * SW_WORDS >= nwords) closed = true;       // nothing behind the next span: nobody else adds to this row
        if (lane == 0 && open_row >= 1 && open_row <= ny) {
            if (closed) y[open_row - 1] = open_sum;
            else atomicAdd(&y[open_row - 1], open_sum);       // the row continues: later waves add their share
        }
    }

    // ---- cells in front of our first semaphore ---------------------------------------------------------------------------
    if (w0 > 0) {
        const bool pbit = (pword >> lane) & 1ull;
        bool covered = __ballot(pbit && pk == SEM_KEY) != 0;
        if (SHARE && wv > 0) covered = sNsem[wv - 1] > 0;          // the span in front is the neighbour's: it counted its semaphores
        for (int64_t w = w0 - 2; (!SHARE || wv == 0) && !covered && w >= 0 && w >= w0 - SW_WORDS; --w) {
            const uint64_t wd = occ[w];
            bool issem = false;
            if ((wd >> lane) & 1ull) issem = keys[(w << 6) + lane] == SEM_KEY;
            covered = __ballot(issem) != 0;
        }
        if (!covered) {
            const int head_end = nsem > 0 ? (int)sList[0] : own_words * 64;
            double sum = 0.0;
            for (int t = lane; t < head_end; t += 64) sum += sP[t];
            sum = wave_reduce_add_f64(sum);
            const int64_t c = carry_in_partition(keys, vals, occ, sems, table_len, (w0 - SW_WORDS > 0 ? w0 - SW_WORDS : 0) << 6);
            if (c >= 1 && c <= table_len) {
                const int64_t row = part_keys[c - 1];
                if (lane == 0 && row >= 1 && row <= ny) atomicAdd(&y[row - 1], sum);
            }
        }
    }
}

template <bool WIDE, bool NT, bool ZFILL, bool SHARE>
__global__ __launch_bounds__(SP_BLOCK) void k_spmv_gather(KeyArr keys, const double* __restrict__ vals,
                                                          const uint64_t* __restrict__ occ, int64_t capacity,
                                                          const int64_t* __restrict__ sems,
                                                          const int64_t* __restrict__ part_keys, int64_t table_len,
                                                          const double* __restrict__ x, int64_t nx,
                                                          double* __restrict__ y, int64_t ny, int pattern) {
    // XCD-aware tile mapping (see k_spmv): XCD g streams the g-th contiguous eighth of the slot array
    const int64_t ntiles = (capacity + SP_TILE - 1) / SP_TILE;
    int64_t tile = blockIdx.x;
    if (ntiles >= 64 && !(pattern & 4)) {
        const int64_t per = (ntiles + 7) / 8;
        tile = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (tile >= ntiles) return;
    }
    gather_tile<WIDE, NT, ZFILL, SHARE>(keys, vals, occ, capacity, sems, part_keys, table_len, x, nx, y, ny, pattern, tile, (int)threadIdx.x);
}

// (the sparse-x product driven by x's stored entries lives in sparsex.hip since round 6: k_spx_accum)

// dense form of a sparse x on the device: xd[j] = x_j, xf[j] = 1 for every STORED entry (explicit zeros included: the
// touched-row pattern of _mul counts stored entries, src/operations.jl:101); both vectors zeroed first
__global__ void k_scatter_x(const int64_t* __restrict__ xi, const double* __restrict__ xv, int64_t nx, double* __restrict__ xd,
                            double* __restrict__ xf, int64_t nxd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx) return;
    const int64_t j = xi[i];
    if (j >= 1 && j <= nxd) { xd[j - 1] = xv[i]; xf[j - 1] = 1.0; }
}
hipError_t launch_scatter_x(const int64_t* xi, const double* xv, int64_t nx, double* xd, double* xf, int64_t nxd, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(xd, 0, (size_t)nxd * sizeof(double), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(xf, 0, (size_t)nxd * sizeof(double), stream);
    if (e != hipSuccess) return e;
    if (nx > 0) hipLaunchKernelGGL(k_scatter_x, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, stream, xi, xv, nx, xd, xf, nxd);
    return hipGetLastError();
}

#include <algorithm>
#include <cstdlib>
// ---- what the gather launch may assume about one orientation (host cache: SpmvMeta in dsa_host.hip) --------------------
// out[0] = longest partition extent in slots (semaphore .. slot in front of the next semaphore / end of the array),
// out[1] = largest difference of two consecutive partition keys, out[2] = first key, out[3] = last key,
// out[4] = 1 if keys do not strictly ascend with the id or a semaphore is missing (tombstone).  Tables without tombstones only.
// One launch, no memset, no atomics on the results: a grid-stride pass with one partial result per workgroup (round 2 issued up to
// three atomicMax per WAVE on the same three words: 359 us for a 1 M-entry table); the workgroup that finishes last (ticket) folds the
// partials into out[0..4] and resets the ticket for the next launch.  scratch = 3 * SPMV_META_BLOCKS partials + the ticket word.
__global__ __launch_bounds__(256) void k_spmv_meta(const int64_t* __restrict__ sems, const int64_t* __restrict__ part_keys,
                                                   int64_t table_len, int64_t capacity, unsigned long long* __restrict__ scratch,
                                                   unsigned long long* __restrict__ out, unsigned long long seq) {
    __shared__ unsigned long long sE[4], sG[4], sB[4];
    __shared__ unsigned int sLast;
    unsigned long long ext = 0, gap = 0, bad = 0;
    // chunks of 1024 entries per workgroup and step, the 16 loads of a thread's four entries issued together (indices clamped, not
    // predicated): one memory round trip per step instead of four
    constexpr int U = 4;
    for (int64_t c = (int64_t)blockIdx.x * (256 * U); c < table_len; c += (int64_t)gridDim.x * (256 * U)) {
        int64_t s0[U], s1[U], k0[U], k1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = c + u * 256 + threadIdx.x;
            const int64_t ic = i < table_len ? i : table_len - 1, in = ic + 1 < table_len ? ic + 1 : ic;
            s0[u] = sems[ic]; s1[u] = sems[in]; k0[u] = part_keys[ic]; k1[u] = part_keys[in];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = c + u * 256 + threadIdx.x;
            if (i >= table_len) continue;
            const bool last = i + 1 >= table_len;
            const int64_t e1 = last ? capacity + 1 : s1[u];
            if (s0[u] <= 0 || e1 <= s0[u]) bad = 1;
            else { const unsigned long long e = (unsigned long long)(e1 - s0[u]); ext = e > ext ? e : ext; }
            if (!last) {
                if (k1[u] <= k0[u]) bad = 1;
                else { const unsigned long long g = (unsigned long long)(k1[u] - k0[u]); gap = g > gap ? g : gap; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long e2 = __shfl_xor(ext, o, 64), g2 = __shfl_xor(gap, o, 64), b2 = __shfl_xor(bad, o, 64);
        ext = e2 > ext ? e2 : ext; gap = g2 > gap ? g2 : gap; bad |= b2;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sE[wv] = ext; sG[wv] = gap; sB[wv] = bad; }
    // the first and the last key of the result: requested by everybody's thread 0 now rather than by the last workgroup at the very end
    const int64_t key_first = threadIdx.x == 0 ? part_keys[0] : 0, key_last = threadIdx.x == 0 ? part_keys[table_len - 1] : 0;
    __syncthreads();
    unsigned long long* ticket = scratch + 3 * SPMV_META_BLOCKS;
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { ext = sE[w] > ext ? sE[w] : ext; gap = sG[w] > gap ? sG[w] : gap; bad |= sB[w]; }
        __hip_atomic_store(scratch + 3 * blockIdx.x + 0, ext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(scratch + 3 * blockIdx.x + 1, gap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(scratch + 3 * blockIdx.x + 2, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // device-scope (write-through) stores, acknowledged before the ticket is taken: a release FENCE here would write back this
        // XCD's whole L2 once per workgroup (cf. k_plan)
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sLast = (t == (unsigned long long)gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!sLast) return;
    // the last workgroup: fold the partials (every other workgroup's stores were acknowledged at device scope before it took its ticket)
    ext = 0; gap = 0; bad = 0;
    {   // all partials of a thread requested in one round (device-scope loads cross to another XCD's data: ~2 us per dependent round)
        constexpr int R = SPMV_META_BLOCKS / 256;
        unsigned long long pe[R], pg[R], pb[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int b = threadIdx.x + r * 256;
            const int bc = b < (int)gridDim.x ? b : 0;
            pe[r] = __hip_atomic_load(scratch + 3 * bc + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pg[r] = __hip_atomic_load(scratch + 3 * bc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pb[r] = __hip_atomic_load(scratch + 3 * bc + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { ext = pe[r] > ext ? pe[r] : ext; gap = pg[r] > gap ? pg[r] : gap; bad |= pb[r]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long e2 = __shfl_xor(ext, o, 64), g2 = __shfl_xor(gap, o, 64), b2 = __shfl_xor(bad, o, 64);
        ext = e2 > ext ? e2 : ext; gap = g2 > gap ? g2 : gap; bad |= b2;
    }
    __syncthreads();
    if (lane == 0) { sE[wv] = ext; sG[wv] = gap; sB[wv] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { ext = sE[w] > ext ? sE[w] : ext; gap = sG[w] > gap ? sG[w] : gap; bad |= sB[w]; }
        // `out` is PINNED HOST memory: the five words, a system-scope release fence, then the sequence number the host polls for —
        // no copy command, no event, no driver call on the host side of the hand-over
        const unsigned long long r[5] = {ext, gap, (unsigned long long)key_first, (unsigned long long)key_last, bad ? 1ull : 0ull};
        for (int q = 0; q < 5; ++q) __hip_atomic_store(out + q, r[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_store(out + 5, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
hipError_t launch_spmv_meta(const int64_t* sems, const int64_t* part_keys, int64_t table_len, int64_t capacity,
                            unsigned long long* scratch, unsigned long long* out6_pinned, unsigned long long seq, hipStream_t stream) {
    if (table_len <= 0) return hipErrorInvalidValue;
    // four entries per thread and step; beyond 256 workgroups the count grows with the table only slowly (1 M entries: 256 workgroups
    // x 4 steps 12-15 us, 977 x 1 step 18-20 us, 64 x 16 steps 26-33 us: every workgroup costs a partial + a ticket at device scope)
    int64_t blocks = (table_len + 1023) / 1024;
    if (blocks > 256) blocks = std::max<int64_t>(256, std::min<int64_t>(SPMV_META_BLOCKS, table_len >> 12));
    { static const char* e = dev_env("DSA_META_BLOCKS"); if (e && atoi(e) > 0 && blocks > atoi(e)) blocks = atoi(e); }
    hipLaunchKernelGGL(k_spmv_meta, dim3((unsigned)blocks), dim3(256), 0, stream, sems, part_keys, table_len, capacity, scratch, out6_pinned, seq);
    return hipGetLastError();
}

template <bool WIDE, bool NT, bool ZFILL>
static void launch_gather_t(int64_t grid, hipStream_t stream, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                            const int64_t* sems, const int64_t* part_keys, int64_t table_len, const double* x, int64_t nx, double* y,
                            int64_t ny, int pattern) {
    // SHARE needs every wave of every workgroup at its barrier: whole tiles only (DSA_SPMV_SHARE=0: the form without the barrier)
    static const bool share_ok = [] { const char* e = dev_env("DSA_SPMV_SHARE"); return !(e && e[0] == '0'); }();
    // SHARE pays where the gathers miss (config 3: 118.1 vs 121.2 us without it); where x is L2-resident the barrier costs more than the
    // ninth word saves (banded shape: 90.9 vs 86.9 us), so the plain-stream instantiations run without it
    const bool share = NT && share_ok && capacity >= SP_TILE && capacity % SP_TILE == 0;
    if (share)
        hipLaunchKernelGGL((k_spmv_gather<WIDE, NT, ZFILL, true>), dim3((unsigned)grid), dim3(SP_BLOCK), 0, stream, keys, vals, occ, capacity, sems,
                           part_keys, table_len, x, nx, y, ny, pattern);
    else
        hipLaunchKernelGGL((k_spmv_gather<WIDE, NT, ZFILL, false>), dim3((unsigned)grid), dim3(SP_BLOCK), 0, stream, keys, vals, occ, capacity, sems,
                           part_keys, table_len, x, nx, y, ny, pattern);
}

// mode bit 0 (SPMV_ZFILL): skip the memset of y, the kernel zeroes the rows without a partition itself (see k_spmv_gather);
// mode bit 1 (SPMV_PLAIN_STREAM): plain instead of non-temporal slot loads.  Scatter form: mode ignored.
static hipError_t launch_spmv(bool scatter, int pattern, int mode, KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                              const int64_t* sems, const int64_t* part_keys, int64_t table_len, const double* x, int64_t nx,
                              double* y, int64_t ny, hipStream_t stream) {
    { static const char* dbg = dev_env("DSA_DBG_SPMV"); if (dbg && pattern == 0) pattern = atoi(dbg) & 4; }
    {   // dev knobs: DSA_SPMV_ZFILL=0 keeps the memset, DSA_SPMV_STREAM=nt|plain forces the stream policy
        static const char* z = dev_env("DSA_SPMV_ZFILL"); if (z && z[0] == '0') mode &= ~1;
        static const char* st = dev_env("DSA_SPMV_STREAM"); if (st) mode = (mode & ~2) | (st[0] == 'p' ? 2 : 0);
    }
    if (scatter) mode = 0;
    const bool zfill = (mode & 1) && table_len > 0;
    if (!zfill) {
        hipError_t e = hipMemsetAsync(y, 0, (size_t)ny * sizeof(double), stream);
        if (e != hipSuccess) return e;
    }
    const int64_t ntiles = (capacity + SP_TILE - 1) / SP_TILE;
    const int64_t grid = ntiles >= 64 ? 8 * ((ntiles + 7) / 8) : ntiles;     // see the XCD-aware mapping in k_spmv
    if (scatter) {
        hipLaunchKernelGGL(k_spmv_scatter, dim3((unsigned)grid), dim3(SP_BLOCK), 0, stream, keys, vals, occ, capacity, sems,
                           part_keys, table_len, x, nx, y, ny, 0);
        return hipGetLastError();
    }
    const int sel = (keys.wide ? 4 : 0) | ((mode & 2) ? 0 : 2) | (zfill ? 1 : 0);
#define DSA_GATHER_CASE(W_, N_, Z_) launch_gather_t<W_, N_, Z_>(grid, stream, keys, vals, occ, capacity, sems, part_keys, table_len, x, nx, y, ny, pattern)
    switch (sel) {
        case 0: DSA_GATHER_CASE(false, false, false); break;
        case 1: DSA_GATHER_CASE(false, false, true); break;
        case 2: DSA_GATHER_CASE(false, true, false); break;
        case 3: DSA_GATHER_CASE(false, true, true); break;
        case 4: DSA_GATHER_CASE(true, false, false); break;
        case 5: DSA_GATHER_CASE(true, false, true); break;
        case 6: DSA_GATHER_CASE(true, true, false); break;
        default: DSA_GATHER_CASE(true, true, true); break;
    }
#undef DSA_GATHER_CASE
    return hipGetLastError();
}

hipError_t launch_spmv_gather(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                              const int64_t* sems, const int64_t* part_keys, const uint8_t*, int64_t table_len,
                              const double* x, int64_t nx, double* y, int64_t ny, int pattern, int mode, hipStream_t stream) {
    return launch_spmv(false, pattern, mode, keys, vals, occ, capacity, sems, part_keys, table_len, x, nx, y, ny, stream);
}
hipError_t launch_spmv_scatter(KeyArr keys, const double* vals, const uint64_t* occ, int64_t capacity,
                               const int64_t* sems, const int64_t* part_keys, const uint8_t*, int64_t table_len,
                               const double* x, int64_t nx, double* y, int64_t ny, hipStream_t stream) {
    return launch_spmv(true, 0, 0, keys, vals, occ, capacity, sems, part_keys, table_len, x, nx, y, ny, stream);
}

}  // namespace dsa
