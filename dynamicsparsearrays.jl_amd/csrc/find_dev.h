// csrc/find_dev.h — device helpers shared by the write sequencer (sequencer.hip) and the batch-parallel insert path
// (parbatch.hip): bitmap scans and the two forms of K-find.  Positions are 1-based like the reference.
#pragma once
#include "dsa_dev.h"

namespace dsa {

static __device__ __forceinline__ uint64_t word_range_mask(int64_t w, int64_t lo0, int64_t hi0) {
    const int64_t b = w << 6;
    int64_t a = lo0 - b, z = hi0 - b;
    if (z < 0 || a > 63) return 0ull;
    if (a < 0) a = 0;
    if (z > 63) z = 63;
    const uint64_t upto = (z == 63) ? ~0ull : ((1ull << (z + 1)) - 1ull);
    return upto & ~mask_lt((int)a);
}

// ---- bitmap scans (uniform, executed by every thread) -------------------------------------------
// _nextemptypos(array, from)  src/utils.jl:3-10
static __device__ int64_t d_next_empty(const uint64_t* occ, int64_t from, int64_t capacity) {
    if (from + 1 > capacity) return 0;
    const int64_t i = from;                 // 0-based index of position from+1
    int64_t w = i >> 6;
    uint64_t word = ~occ[w] & ~mask_lt((int)(i & 63));
    const int64_t lastw = (capacity - 1) >> 6;
    while (true) {
        if (word) {
            const int64_t p = (w << 6) + __ffsll((unsigned long long)word);   // 1-based position
            return p <= capacity ? p : 0;
        }
        if (++w > lastw) return 0;
        word = ~occ[w];
    }
}
// _previousemptypos(array, from)  src/utils.jl:21-28
static __device__ int64_t d_prev_empty(const uint64_t* occ, int64_t from) {
    if (from - 1 < 1) return 0;
    const int64_t i = from - 2;             // 0-based index of position from-1
    int64_t w = i >> 6;
    const int b = (int)(i & 63);
    uint64_t word = ~occ[w] & (b == 63 ? ~0ull : mask_lt(b + 1));
    while (true) {
        if (word) return (w << 6) + (63 - __clzll((long long)word)) + 1;
        if (--w < 0) return 0;
        word = ~occ[w];
    }
}
// largest occupied position in [lo, pos], or lo-1  (the walk-left loops of src/finds.jl:33-35,50-52)
static __device__ int64_t d_prev_occupied(const uint64_t* occ, int64_t pos, int64_t lo) {
    if (pos < lo) return lo - 1;
    const int64_t i = pos - 1;
    int64_t w = i >> 6;
    const int b = (int)(i & 63);
    uint64_t word = occ[w] & (b == 63 ? ~0ull : mask_lt(b + 1));
    const int64_t low = (lo - 1) >> 6;
    while (true) {
        if (word) {
            const int64_t p = (w << 6) + (63 - __clzll((long long)word)) + 1;
            return p >= lo ? p : lo - 1;
        }
        if (--w < low) return lo - 1;
        word = occ[w];
    }
}

struct DFound { int64_t pos; int64_t key; double val; bool has; };

// find(array, key, from, to)  src/finds.jl:29-57 — same probes, same answers
static __device__ DFound d_find(KeyArr keys, const double* vals, const uint64_t* occ, int64_t key, int64_t from, int64_t to) {
    while (from <= to) {
        const int64_t mid = (from + to) >> 1;
        const int64_t i = d_prev_occupied(occ, mid, from);
        if (i < from) {
            from = mid + 1;
        } else {
            const int64_t ck = keys[i - 1];
            if (ck > key) to = i - 1;
            else if (ck < key) from = mid + 1;
            else return DFound{i, ck, vals[i - 1], true};
        }
    }
    const int64_t i = to >= 1 ? d_prev_occupied(occ, to, 1) : 0;
    if (i > 0) return DFound{i, keys[i - 1], vals[i - 1], true};
    return DFound{0, 0, 0.0, false};
}

// smallest occupied position in (pos, hi], or 0
static __device__ int64_t d_next_occupied(const uint64_t* occ, int64_t pos, int64_t hi) {
    if (pos + 1 > hi) return 0;
    const int64_t i = pos;                  // 0-based index of position pos+1
    int64_t w = i >> 6;
    uint64_t word = occ[w] & ~mask_lt((int)(i & 63));
    const int64_t lastw = (hi - 1) >> 6;
    while (true) {
        if (word) {
            const int64_t p = (w << 6) + __ffsll((unsigned long long)word);
            return p <= hi ? p : 0;
        }
        if (++w > lastw) return 0;
        word = occ[w];
    }
}

// K-find, wave-parallel form.  Same answer as d_find / the reference bisection whenever, inside
// [from, to], every occupied key < `key` precedes every occupied key > `key` (true for a vector's PMA, for
// a partition without its semaphore, and for a partition WITH its semaphore when key > 0).  Then find()
// returns: the cell holding `key` if present; else the last cell of the range with a smaller key; else the
// nearest occupied cell left of `from`; else (0, nothing)  (src/finds.jl:29-57).  A 64-ary search on slot
// positions: each round every lane probes one position (nearest occupied cell at or before it, via the
// bitmap), one ballot narrows the interval 64-fold: ~log64(range) dependent round trips instead of log2.
static __device__ DFound d_find_fast(KeyArr keys, const double* vals, const uint64_t* occ, int64_t key, int64_t from, int64_t to) {
    const int lane = lane_id();
    int64_t pstar;
    if (to < from) {
        pstar = to;
    } else {
        int64_t L = from - 1, H = to;        // invariant: every occupied cell of [from, L] has a key < `key`; the boundary lies in [L, H]
        while (H - L > 64) {
            const int64_t width = H - L;
            const int64_t p = L + (width * (lane + 1)) / 64;
            const int64_t q = d_prev_occupied(occ, p, L + 1);
            bool pr = true;
            if (q > L) pr = keys[q - 1] < key;
            const uint64_t nb = ~__ballot(pr);
            const int j = nb ? __ffsll((unsigned long long)nb) - 1 : 64;      // lanes 0..j-1 true, lane j false
            const int64_t pj = L + (width * (j + 1)) / 64;
            const int64_t pj1 = L + (width * j) / 64;
            if (j < 64) H = pj - 1;
            L = pj1;
        }
        const int64_t p = L + 1 + lane;
        bool viol = false;
        if (p <= H && occ_test(occ, p)) viol = keys[p - 1] >= key;
        const uint64_t b = __ballot(viol);
        pstar = b ? L + __ffsll((unsigned long long)b) - 1 : H;
        const int64_t nxt = d_next_occupied(occ, pstar, to);
        if (nxt != 0 && keys[nxt - 1] == key) return DFound{nxt, key, vals[nxt - 1], true};
    }
    const int64_t i = pstar >= 1 ? d_prev_occupied(occ, pstar, 1) : 0;      // may lie left of `from`, like the reference
    if (i > 0) return DFound{i, keys[i - 1], vals[i - 1], true};
    return DFound{0, 0, 0.0, false};
}


struct DFoundKey { int64_t pos; int64_t key; bool has; };
// find(col_keys, key)  src/finds.jl:59-61 on a Vector{Union{Nothing,L}}
static __device__ DFoundKey d_find_table(const int64_t* ck, const uint8_t* live, int64_t len, int64_t key) {
    int64_t from = 1, to = len;
    while (from <= to) {
        const int64_t mid = (from + to) >> 1;
        int64_t i = mid;
        while (i >= from && !live[i - 1]) --i;
        if (i < from) {
            from = mid + 1;
        } else {
            const int64_t c = ck[i - 1];
            if (c > key) to = i - 1;
            else if (c < key) from = mid + 1;
            else return DFoundKey{i, c, true};
        }
    }
    int64_t i = to;
    while (i > 0 && !live[i - 1]) --i;
    if (i > 0) return DFoundKey{i, ck[i - 1], true};
    return DFoundKey{0, 0, false};
}

// wave-parallel form of d_find_table (live column keys are strictly ascending)
// dense = true: the caller knows that no entry of [0, len) is tombstoned (nb_partitions == table_len): live[] is not read
static __device__ DFoundKey d_find_table_fast(const int64_t* ck, const uint8_t* live, int64_t len, int64_t key, bool dense = false) {
    const int lane = lane_id();
    int64_t L = 0, H = len;
    while (H - L > 64) {
        const int64_t width = H - L;
        const int64_t p = L + (width * (lane + 1)) / 64;
        int64_t q = p;
        if (!dense) while (q > L && !live[q - 1]) --q;
        bool pr = true;
        if (q > L) pr = ck[q - 1] < key;
        const uint64_t nb = ~__ballot(pr);
        const int j = nb ? __ffsll((unsigned long long)nb) - 1 : 64;
        const int64_t pj = L + (width * (j + 1)) / 64;
        const int64_t pj1 = L + (width * j) / 64;
        if (j < 64) H = pj - 1;
        L = pj1;
    }
    const int64_t p = L + 1 + lane;
    bool viol = false;
    int64_t myk = 0;
    if (p <= H && (dense || live[p - 1])) { myk = ck[p - 1]; viol = myk >= key; }
    const uint64_t b = __ballot(viol);
    const int64_t pstar = b ? L + __ffsll((unsigned long long)b) - 1 : H;
    if (dense) {
        // the two keys the answer needs were just loaded by the lanes that probed positions pstar + 1 and pstar (if they lie in
        // (L, H]): a shuffle instead of two more dependent round trips
        const int64_t nxt = pstar + 1;
        if (nxt <= len) {
            const int64_t kn = (nxt > L && nxt <= H) ? __shfl(myk, (int)(nxt - L - 1), 64) : ck[nxt - 1];
            if (kn == key) return DFoundKey{nxt, key, true};
        }
        if (pstar > 0) {
            const int64_t ki = (pstar > L && pstar <= H) ? __shfl(myk, (int)(pstar - L - 1), 64) : ck[pstar - 1];
            return DFoundKey{pstar, ki, true};
        }
        return DFoundKey{0, 0, false};
    }
    int64_t nxt = pstar + 1;
    while (nxt <= len && !live[nxt - 1]) ++nxt;
    if (nxt <= len && ck[nxt - 1] == key) return DFoundKey{nxt, key, true};
    int64_t i = pstar;
    while (i > 0 && !live[i - 1]) --i;
    if (i > 0) return DFoundKey{i, ck[i - 1], true};
    return DFoundKey{0, 0, false};
}

// _nextnonemptypos(semaphores, from)  src/utils.jl:12-19
static __device__ int64_t d_next_live_sem(const int64_t* sems, int64_t from, int64_t len) {
    int64_t pos = from + 1;
    while (pos <= len) {
        if (sems[pos - 1] != 0) return pos;
        ++pos;
    }
    return 0;
}

}  // namespace dsa
