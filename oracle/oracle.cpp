// oracle/oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see oracle.hpp).
//
// Restates, function by function, the reference algorithm
// (atoptima/DynamicSparseArrays.jl v0.7.2).  Compile with
//   g++ -std=c++17 -O2 -ffp-contract=off
// so that every Float64 expression is evaluated exactly as Julia evaluates it
// (separate IEEE multiply / add / divide, no FMA contraction).
#include "oracle.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace ora {

[[noreturn]] static void fail(int32_t code, const std::string& msg) { throw Err{code, msg}; }
#define ORA_ASSERT(cond, what) do { if (!(cond)) fail(EASSERT, what); } while (0)

// ---------------------------------------------------------------------------
// src/utils.jl
// ---------------------------------------------------------------------------
// _nextemptypos   src/utils.jl:3-10
int64_t nextemptypos(const Elements& a, int64_t from) {
    int64_t pos = from + 1;
    while (pos <= a.len()) {
        if (a.empty_at(pos)) return pos;
        pos += 1;
    }
    return 0;
}
// _nextnonemptypos   src/utils.jl:12-19
int64_t nextnonemptypos(const Elements& a, int64_t from) {
    int64_t pos = from + 1;
    while (pos <= a.len()) {
        if (!a.empty_at(pos)) return pos;
        pos += 1;
    }
    return 0;
}
int64_t nextnonemptypos(const Table& a, int64_t from) {
    int64_t pos = from + 1;
    while (pos <= a.len()) {
        if (!a.empty_at(pos)) return pos;
        pos += 1;
    }
    return 0;
}
// _previousemptypos   src/utils.jl:21-28
int64_t previousemptypos(const Elements& a, int64_t from) {
    int64_t pos = from - 1;
    while (pos >= 1) {
        if (a.empty_at(pos)) return pos;
        pos -= 1;
    }
    return 0;
}
// _nbcells (from included, to excluded)   src/utils.jl:48-58
int64_t nbcells(const Elements& a, int64_t from, int64_t to) {
    ORA_ASSERT(1 <= from && to <= a.len() + 1, "_nbcells range");
    if (from >= to) return 0;
    int64_t nb = 0;
    for (int64_t pos = from; pos <= to - 1; ++pos)
        if (!a.empty_at(pos)) nb += 1;
    return nb;
}

// ---------------------------------------------------------------------------
// src/moves.jl
// ---------------------------------------------------------------------------
static const int64_t SEM_KEY = 0;   // semaphore_key(::Type{<:Integer})   src/pcsr.jl:23

static void check_move_args(const Elements& a, int64_t from, int64_t to) {
    // src/moves.jl:8-11 / :51-54 — `array[to]` is evaluated first (BoundsError
    // if `to` is outside), then the ArgumentError, then the explicit bounds.
    const int64_t len = a.len();
    if (!(1 <= to && to <= len)) fail(EBOUNDS, "cannot access array at index [to]");
    if (!a.empty_at(to)) fail(EARG, "The cell erased by the movement must contain nothing.");
    if (!(1 <= from && from <= len)) fail(EBOUNDS, "cannot access array at index [from]");
}

// _movecellstoright! + _moverightloop!   src/moves.jl:7-42
void movecellstoright(Elements& a, int64_t from, int64_t to, Table* sem) {
    check_move_args(a, from, to);
    int64_t i = to;
    while (i > from) {
        i -= 1;
        const bool occ = !a.empty_at(i);
        if (occ) {
            const Cell c = a.cell[i - 1];
            if (sem && c.key == SEM_KEY) sem->v[(int64_t)c.val - 1] = i + 1;
            a.set(i + 1, c.key, c.val);
        } else {
            a.clear(i + 1);
        }
    }
    a.clear(i);
}

// _movecellstoleft! + _moveleftloop!   src/moves.jl:50-85
void movecellstoleft(Elements& a, int64_t from, int64_t to, Table* sem) {
    check_move_args(a, from, to);
    int64_t i = to;
    while (i < from) {
        i += 1;
        const bool occ = !a.empty_at(i);
        if (occ) {
            const Cell c = a.cell[i - 1];
            if (sem && c.key == SEM_KEY) sem->v[(int64_t)c.val - 1] = i - 1;
            a.set(i - 1, c.key, c.val);
        } else {
            a.clear(i - 1);
        }
    }
    a.clear(i);
}

// pack!   src/moves.jl:94-110
void pack(Elements& a, int64_t ws, int64_t /*we*/, int64_t m) {
    int64_t i = ws, j = ws;
    while (i < ws + m) {
        if (a.empty_at(j)) { j += 1; continue; }
        if (i < j) {
            a.set(i, a.cell[j - 1].key, a.cell[j - 1].val);
            a.clear(j);
        }
        i += 1;
        j += 1;
    }
}

// spread!(array, ws, we, m)   src/moves.jl:120-140
void spread(Elements& a, int64_t ws, int64_t we, int64_t m) {
    const int64_t capacity = we - ws + 1;
    int64_t nb_empty_cells = capacity - m;
    const double empty_cell_freq = (double)capacity / (double)nb_empty_cells;   // Int/Int -> Float64
    double next_empty_cell = ((double)ws + std::floor((double)nb_empty_cells * empty_cell_freq)) - 1.0;
    int64_t i = ws + m - 1;
    int64_t j = we;
    while (i != j && i >= ws) {
        if ((double)j == next_empty_cell) {
            nb_empty_cells -= 1;
            next_empty_cell = ((double)ws + std::floor((double)nb_empty_cells * empty_cell_freq)) - 1.0;
            j -= 1;
        } else {
            a.set(j, a.cell[i - 1].key, a.cell[i - 1].val);
            a.clear(i);
            i -= 1;
            j -= 1;
        }
    }
}

// spread!(array, ws, we, m, semaphores)   src/moves.jl:142-171
void spread_sem(Elements& a, int64_t ws, int64_t we, int64_t m, Table* sem) {
    const int64_t capacity = we - ws + 1;
    int64_t nb_empty_cells = capacity - m;
    const double empty_cell_freq = (double)capacity / (double)nb_empty_cells;
    double next_empty_cell = ((double)ws + std::floor((double)nb_empty_cells * empty_cell_freq)) - 1.0;
    int64_t i = ws + m - 1;
    int64_t j = we;
    while (i >= ws) {
        if ((double)j == next_empty_cell) {
            nb_empty_cells -= 1;
            next_empty_cell = ((double)ws + std::floor((double)nb_empty_cells * empty_cell_freq)) - 1.0;
            j -= 1;
        } else {
            if (i != j) {
                a.set(j, a.cell[i - 1].key, a.cell[i - 1].val);
                a.clear(i);
            }
            if (sem && !a.empty_at(j)) {
                const Cell c = a.cell[j - 1];
                if (c.key == SEM_KEY) sem->v[(int64_t)c.val - 1] = j;
            }
            i -= 1;
            j -= 1;
        }
    }
}

// ---------------------------------------------------------------------------
// src/finds.jl
// ---------------------------------------------------------------------------
// find(array, key, from, to)   src/finds.jl:29-57
Found find(const Elements& a, int64_t key, int64_t from, int64_t to) {
    while (from <= to) {
        const int64_t mid = (from + to) / 2;   // positions are positive: ÷ == /
        int64_t i = mid;
        while (i >= from && a.empty_at(i)) i -= 1;
        if (i < from) {
            from = mid + 1;
        } else {
            const int64_t curkey = a.cell[i - 1].key;
            if (curkey > key) to = i - 1;
            else if (curkey < key) from = mid + 1;
            else return Found{i, true, a.cell[i - 1]};
        }
    }
    int64_t i = to;
    while (i > 0 && a.empty_at(i)) i -= 1;
    if (i > 0) return Found{i, true, a.cell[i - 1]};
    return Found{0, false, Cell{0, 0.0}};
}

// find(col_keys, key): the same routine on a Vector{Union{Nothing,L}}
// (src/finds.jl:29-61 with _getkey of src/utils.jl:30-35)
FoundKey find(const Table& a, int64_t key) {
    int64_t from = 1, to = a.len();
    while (from <= to) {
        const int64_t mid = (from + to) / 2;
        int64_t i = mid;
        while (i >= from && a.empty_at(i)) i -= 1;
        if (i < from) {
            from = mid + 1;
        } else {
            const int64_t curkey = a.v[i - 1];
            if (curkey > key) to = i - 1;
            else if (curkey < key) from = mid + 1;
            else return FoundKey{i, true, curkey};
        }
    }
    int64_t i = to;
    while (i > 0 && a.empty_at(i)) i -= 1;
    if (i > 0) return FoundKey{i, true, a.v[i - 1]};
    return FoundKey{0, false, 0};
}

// ---------------------------------------------------------------------------
// src/writes.jl
// ---------------------------------------------------------------------------
// _insert!(array, key, value, pos, semaphores)   src/writes.jl:26-43
InsRes insert_after(Elements& a, int64_t key, double value, int64_t pos, Table* sem) {
    int64_t insertion_pos = pos;
    const int64_t next_empty_pos = nextemptypos(a, pos);
    if (next_empty_pos != 0) {
        movecellstoright(a, pos + 1, next_empty_pos, sem);
        a.set(pos + 1, key, value);
        insertion_pos += 1;
    } else {
        const int64_t previous_empty_pos = previousemptypos(a, pos);
        if (previous_empty_pos != 0) {
            movecellstoleft(a, pos, previous_empty_pos, sem);
            a.set(pos, key, value);
        } else {
            fail(EFULL, "No empty cell to insert a new element.");
        }
    }
    return InsRes{insertion_pos, true};
}

// insert!(array, key, value, from, to, semaphores)   src/writes.jl:14-23
InsRes insert(Elements& a, int64_t key, double value, int64_t from, int64_t to, Table* sem) {
    const Found f = find(a, key, from, to);
    if (f.has && f.elem.key == key && from <= f.pos && f.pos <= to) {
        a.set(f.pos, key, value);
        return InsRes{f.pos, false};
    }
    return insert_after(a, key, value, f.pos, sem);
}

// delete!(array, key, from, to) + _delete!   src/writes.jl:57-68
InsRes erase(Elements& a, int64_t key, int64_t from, int64_t to) {
    const Found f = find(a, key, from, to);
    if (f.has && f.elem.key == key) {
        a.clear(f.pos);
        return InsRes{f.pos, true};
    }
    return InsRes{0, false};
}

// purge!(array, from, to)   src/writes.jl:80-91
PurgeRes purge(Elements& a, int64_t from, int64_t to) {
    if (to < from) return PurgeRes{0, 0};
    int64_t nb = 0;
    for (int64_t pos = from; pos <= to; ++pos) {
        if (!a.empty_at(pos)) { a.clear(pos); nb += 1; }
    }
    const int64_t mid = from + (to - from) / 2;
    return PurgeRes{mid, nb};
}

// ---------------------------------------------------------------------------
// src/pma.jl
// ---------------------------------------------------------------------------
// capacity = 2^ceil(Int, log2(ceil(nb_elements / t_h)))   src/pma.jl:64,81,88
static int64_t capacity_for(int64_t nb_elements, double t_h) {
    const double c = std::ceil((double)nb_elements / t_h);
    const int64_t e = (int64_t)std::ceil(std::log2(c));
    return (int64_t)1 << e;
}

// _pma   src/pma.jl:42-55
static void pma_finish(PMA& p, int64_t nb_elements) {
    const int64_t capacity = p.array.len();
    const double lc = std::log2((double)capacity);
    const int64_t nb_segs = (int64_t)1 << (int64_t)std::ceil(std::log2((double)capacity / lc));
    const int64_t seg_capacity = capacity / nb_segs;
    const int64_t height = (int64_t)std::log2((double)nb_segs);
    p.capacity = capacity;
    p.segment_capacity = seg_capacity;
    p.nb_segments = nb_segs;
    p.nb_elements = nb_elements;
    p.height = height;
    p.t_h = 0.7; p.t_0 = 0.92; p.p_h = 0.3; p.p_0 = 0.08;   // src/pma.jl:58,70,87
    p.t_d = (p.t_h - p.t_0) / (double)height;
    p.p_d = (p.p_h - p.p_0) / (double)height;
    even_rebalance(p, 1, capacity, nb_elements);
}

// PackedMemoryArray(K, T; expected_nb_elems = 100)   src/pma.jl:86-91
void pma_init_empty(PMA& p, int64_t expected_nb_elems) {
    p = PMA();
    const int64_t capacity = capacity_for(expected_nb_elems, 0.7);
    p.array.resize(capacity);
    pma_finish(p, 0);
}

// PackedMemoryArray(keys, values; sort)   src/pma.jl:69-84 (+ _array :34-40)
void pma_init(PMA& p, std::vector<int64_t>& keys, std::vector<double>& vals, bool sort) {
    const int64_t n = (int64_t)vals.size();
    if (n == 0) { pma_init_empty(p); return; }
    if (sort) {   // sortperm (stable) + permute!
        std::vector<int64_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) { return keys[x] < keys[y]; });
        std::vector<int64_t> k2(n); std::vector<double> v2(n);
        for (int64_t i = 0; i < n; ++i) { k2[i] = keys[perm[i]]; v2[i] = vals[perm[i]]; }
        keys.swap(k2); vals.swap(v2);
    }
    p = PMA();
    const int64_t capacity = capacity_for(n, 0.7);
    p.array.resize(capacity);
    for (int64_t i = 1; i <= n; ++i) p.array.set(i, keys[i - 1], vals[i - 1]);
    pma_finish(p, n);
}

// _even_rebalance!(pma, ws, we, m)   src/pma.jl:94-103
void even_rebalance(PMA& p, int64_t ws, int64_t we, int64_t m) {
    const int64_t capacity = we - ws + 1;
    if (capacity == p.segment_capacity) return;
    p.stat_rebalances += 1; p.stat_window_slots += capacity;
    pack(p.array, ws, we, m);
    spread(p.array, ws, we, m);
}

// _extend!   src/pma.jl:143-151
static void extend(PMA& p) {
    p.capacity *= 2;
    p.nb_segments *= 2;
    p.height += 1;
    p.t_d = (p.t_h - p.t_0) / (double)p.height;
    p.p_d = (p.p_h - p.p_0) / (double)p.height;
    p.array.resize(p.capacity);
    p.stat_extends += 1;
}
// _shrink!   src/pma.jl:153-161
static void shrink(PMA& p) {
    p.capacity /= 2;
    p.nb_segments /= 2;
    p.height -= 1;
    p.t_d = (p.t_h - p.t_0) / (double)p.height;
    p.p_d = (p.p_h - p.p_0) / (double)p.height;
    p.array.resize(p.capacity);
    p.stat_shrinks += 1;
}

// _look_for_rebalance!   src/pma.jl:105-141
Window look_for_rebalance(PMA& pma, int64_t pos) {
    double p = 0.0, t = 0.0, density = 0.0;
    int64_t height = 0;
    int64_t prev_win_start = pos;
    int64_t prev_win_end = pos - 1;
    int64_t nb_cells_left = 0, nb_cells_right = 0;
    while (height <= pma.height) {
        const int64_t window_capacity = ((int64_t)1 << height) * pma.segment_capacity;
        const int64_t win_start = ((pos - 1) / window_capacity) * window_capacity + 1;
        const int64_t win_end = win_start + window_capacity - 1;
        nb_cells_left += nbcells(pma.array, win_start, prev_win_start);
        nb_cells_right += nbcells(pma.array, prev_win_end + 1, win_end + 1);
        density = (double)(nb_cells_left + nb_cells_right) / (double)window_capacity;
        p = pma.p_0 + pma.p_d * (double)height;
        t = pma.t_0 + pma.t_d * (double)height;
        if (p <= density && density <= t) {
            return Window{win_start, win_end, nb_cells_left + nb_cells_right};
        }
        prev_win_start = win_start;
        prev_win_end = win_end;
        height += 1;
    }
    const int64_t nb_cells = nb_cells_left + nb_cells_right;
    if (density > t) extend(pma);
    if (density < p && pma.height > 1) {
        // "We must pack before shrinking otherwise we loose data"
        pack(pma.array, 1, pma.array.len() / 2, nb_cells);
        shrink(pma);
    }
    return Window{1, pma.capacity, nb_cells};
}

// getindex(pma, key)   src/pma.jl:185-193
double pma_get(const PMA& p, int64_t key) {
    const Found f = find(p.array, key, 1, p.array.len());
    if (f.has && f.elem.key == key) return f.elem.val;
    return 0.0;
}

// setindex!(pma, value, key)   src/pma.jl:196-213
void pma_set(PMA& p, int64_t key, double value) {
    if (value != 0.0) {
        const InsRes r = insert(p.array, key, value, 1, p.array.len(), nullptr);
        if (r.is_new) {
            p.nb_elements += 1;
            const Window w = look_for_rebalance(p, r.pos);
            even_rebalance(p, w.ws, w.we, w.count);
        }
    } else {
        const InsRes r = erase(p.array, key, 1, p.array.len());
        if (r.is_new) {
            p.nb_elements -= 1;
            const Window w = look_for_rebalance(p, r.pos);
            even_rebalance(p, w.ws, w.we, w.count);
        }
    }
}

// ---------------------------------------------------------------------------
// src/vector.jl
// ---------------------------------------------------------------------------
static double combine_apply(int32_t op, double a, double b) {
    switch (op) {
        case COMBINE_ADD: return a + b;
        case COMBINE_MUL: return a * b;
        default: return b;
    }
}

// _prepare_keys_vals!   src/vector.jl:10-36
void prepare_keys_vals(std::vector<int64_t>& keys, std::vector<double>& vals, int32_t combine) {
    ORA_ASSERT(keys.size() == vals.size(), "length(keys) == length(values)");
    const int64_t n = (int64_t)keys.size();
    if (n == 0) return;
    std::vector<int64_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) { return keys[x] < keys[y]; });
    {
        std::vector<int64_t> k2(n); std::vector<double> v2(n);
        for (int64_t i = 0; i < n; ++i) { k2[i] = keys[perm[i]]; v2[i] = vals[perm[i]]; }
        keys.swap(k2); vals.swap(v2);
    }
    int64_t write_pos = 1, read_pos = 1;
    int64_t prev_id = keys[read_pos - 1];
    while (read_pos < n) {
        read_pos += 1;
        const int64_t cur_id = keys[read_pos - 1];
        if (prev_id == cur_id) {
            vals[write_pos - 1] = combine_apply(combine, vals[write_pos - 1], vals[read_pos - 1]);
        } else {
            write_pos += 1;
            if (write_pos < read_pos) {
                keys[write_pos - 1] = cur_id;
                vals[write_pos - 1] = vals[read_pos - 1];
            }
        }
        prev_id = cur_id;
    }
    keys.resize(write_pos);
    vals.resize(write_pos);
}

// _dynamicsparsevec / dynamicsparsevec   src/vector.jl:38-62 ; n < 0 => _guess_length(I) (:6)
void vec_init(DynVec& v, std::vector<int64_t> keys, std::vector<double> vals, int32_t combine, int64_t n) {
    if (keys.size() != vals.size()) fail(EARG, "keys & nonzeros vectors must have same length.");
    if (n < 0) {
        n = 0;
        for (int64_t k : keys) n = std::max(n, k);
    }
    prepare_keys_vals(keys, vals, combine);
    pma_init(v.pma, keys, vals, true);
    v.n = n;
}

// setindex!(v, value, key)   src/vector.jl:76-81
void vec_set(DynVec& v, int64_t key, double value) {
    if (value != 0.0) v.n = std::max(v.n, key);
    pma_set(v.pma, key, value);
}

// ---------------------------------------------------------------------------
// src/pcsr.jl
// ---------------------------------------------------------------------------
// PackedCSC(K, T)   src/pcsr.jl:65-68
void pcsc_init_empty(PackedCSC& c) {
    c = PackedCSC();
    pma_init_empty(c.pma);
}

// PackedCSC(row_keys, values, combine)   src/pcsr.jl:26-63
void pcsc_init(PackedCSC& c, const std::vector<std::vector<int64_t>>& row_keys,
               const std::vector<std::vector<double>>& values, int32_t combine) {
    c = PackedCSC();
    const int64_t nb_semaphores = (int64_t)row_keys.size();
    ORA_ASSERT(nb_semaphores == (int64_t)values.size(), "nb_semaphores == length(values)");
    std::vector<int64_t> pk; std::vector<double> pv;
    for (int64_t id = 1; id <= nb_semaphores; ++id) {
        pk.push_back(SEM_KEY);
        pv.push_back((double)id);
        std::vector<int64_t> nk = row_keys[id - 1];
        std::vector<double> nv = values[id - 1];
        prepare_keys_vals(nk, nv, combine);
        for (size_t j = 0; j < nk.size(); ++j) { pk.push_back(nk[j]); pv.push_back(nv[j]); }
    }
    pma_init(c.pma, pk, pv, false);
    c.semaphores.resize(nb_semaphores);
    for (int64_t pos = 1; pos <= c.pma.array.len(); ++pos) {
        if (!c.pma.array.empty_at(pos) && c.pma.array.cell[pos - 1].key == SEM_KEY) {
            const int64_t id = (int64_t)c.pma.array.cell[pos - 1].val;
            c.semaphores.v[id - 1] = pos;
            c.semaphores.live[id - 1] = 1;
        }
    }
    c.nb_partitions = nb_semaphores;
}

// _even_rebalance!(pcsc, ...)   src/pcsr.jl:88-97
void pcsc_even_rebalance(PackedCSC& c, int64_t ws, int64_t we, int64_t m) {
    const int64_t capacity = we - ws + 1;
    if (capacity == c.pma.segment_capacity) return;
    c.pma.stat_rebalances += 1; c.pma.stat_window_slots += capacity;
    pack(c.pma.array, ws, we, m);
    spread_sem(c.pma.array, ws, we, m, &c.semaphores);
}

// addpartition!(pcsc)   src/pcsr.jl:99-112
void addpartition(PackedCSC& c) {
    const int64_t sem_pos = c.pma.array.len();
    c.nb_partitions += 1;
    c.semaphores.push(sem_pos);
    const double sem_val = (double)c.semaphores.len();
    const InsRes r = insert_after(c.pma.array, SEM_KEY, sem_val, sem_pos, &c.semaphores);
    if (r.is_new) {
        c.pma.nb_elements += 1;
        const Window w = look_for_rebalance(c.pma, r.pos);
        pcsc_even_rebalance(c, w.ws, w.we, w.count);
    }
}

// addpartition!(pcsc, prev_sem_id)   src/pcsr.jl:114-146
void addpartition(PackedCSC& c, int64_t prev_sem_id) {
    Table& semaphores = c.semaphores;
    const int64_t nb_semaphores = semaphores.len();
    int64_t sem_pos = 0;
    if (prev_sem_id + 1 < 1 || prev_sem_id + 1 > nb_semaphores) fail(EBOUNDS, "semaphores[prev_sem_id + 1]");
    if (semaphores.empty_at(prev_sem_id + 1)) {
        const int64_t next_sem_id = nextnonemptypos(semaphores, prev_sem_id + 1);
        // the reference indexes semaphores[next_sem_id]; next_sem_id == 0 => BoundsError
        if (next_sem_id == 0) fail(EBOUNDS, "semaphores[0] (no live partition after a tombstone)");
        sem_pos = semaphores.v[next_sem_id - 1] - 1;
        semaphores.live[prev_sem_id] = 1;   // slot is (re)filled below (:138)
    } else {
        sem_pos = semaphores.v[prev_sem_id] - 1;
        semaphores.resize(nb_semaphores + 1);
        for (int64_t i = nb_semaphores; i >= prev_sem_id + 1; --i) {
            const bool live = !semaphores.empty_at(i);
            const int64_t moved_sem_pos = semaphores.v[i - 1];
            semaphores.v[i] = semaphores.v[i - 1];
            semaphores.live[i] = semaphores.live[i - 1];
            ORA_ASSERT(live, "!isnothing(moved_sem_pos)");   // src/pcsr.jl:132
            c.pma.array.set(moved_sem_pos, SEM_KEY, (double)(i + 1));
        }
    }
    c.nb_partitions += 1;
    const double sem_val = (double)(prev_sem_id + 1);
    const InsRes r = insert_after(c.pma.array, SEM_KEY, sem_val, sem_pos, &c.semaphores);
    semaphores.v[prev_sem_id] = r.pos;
    semaphores.live[prev_sem_id] = 1;
    if (r.is_new) {
        c.pma.nb_elements += 1;
        const Window w = look_for_rebalance(c.pma, r.pos);
        pcsc_even_rebalance(c, w.ws, w.we, w.count);
    }
}

// addcolumn!   src/pcsr.jl:148-169
static int64_t addcolumn(MappedPackedCSC& m, int64_t col, int64_t prev_col_pos) {
    int64_t col_pos = 0;
    if (prev_col_pos == m.col_keys.len()) {
        m.col_keys.push(col);
        addpartition(m.pcsc);
        col_pos = m.col_keys.len();
    } else {
        if (m.col_keys.empty_at(prev_col_pos + 1)) {
            m.col_keys.v[prev_col_pos] = col;
            m.col_keys.live[prev_col_pos] = 1;
        } else {
            const int64_t nbcolkeys = m.col_keys.len();
            m.col_keys.resize(nbcolkeys + 1);
            for (int64_t i = nbcolkeys; i >= prev_col_pos + 1; --i) {
                m.col_keys.v[i] = m.col_keys.v[i - 1];
                m.col_keys.live[i] = m.col_keys.live[i - 1];
            }
            m.col_keys.v[prev_col_pos] = col;
            m.col_keys.live[prev_col_pos] = 1;
        }
        addpartition(m.pcsc, prev_col_pos);
        col_pos = prev_col_pos + 1;
    }
    return col_pos;
}

// _pos_of_partition_start   src/pcsr.jl:171-175
int64_t pos_of_partition_start(const PackedCSC& c, int64_t partition) {
    if (partition < 1 || partition > c.semaphores.len()) fail(EBOUNDS, "semaphores[partition]");
    ORA_ASSERT(!c.semaphores.empty_at(partition), "!isnothing(partition_start_pos)");
    return c.semaphores.v[partition - 1];
}
// _pos_of_partition_end   src/pcsr.jl:177-186
int64_t pos_of_partition_end(const PackedCSC& c, int64_t partition) {
    int64_t pos = c.pma.array.len();
    const int64_t next_partition = nextnonemptypos(c.semaphores, partition);
    if (next_partition != 0) pos = c.semaphores.v[next_partition - 1] - 1;
    return pos;
}

// deletepartition!   src/pcsr.jl:188-204
void deletepartition(PackedCSC& c, int64_t partition) {
    const int64_t len = c.semaphores.len();
    if (!(1 <= partition && partition <= len)) fail(EBOUNDS, "cannot access partition at index");
    c.nb_partitions -= 1;
    const int64_t sem_pos = pos_of_partition_start(c, partition);
    const int64_t partition_end_pos = pos_of_partition_end(c, partition);
    const PurgeRes pr = purge(c.pma.array, sem_pos, partition_end_pos);
    if (pr.nb > 0) {
        c.pma.nb_elements -= pr.nb;
        const Window w = look_for_rebalance(c.pma, pr.mid);
        pcsc_even_rebalance(c, w.ws, w.we, w.count);
    }
    c.semaphores.live[partition - 1] = 0;
    c.semaphores.v[partition - 1] = 0;
}

// getindex(pcsc, key, partition)   src/pcsr.jl:222-232
double pcsc_get(const PackedCSC& c, int64_t key, int64_t partition) {
    const int64_t from = pos_of_partition_start(c, partition);
    const int64_t to = pos_of_partition_end(c, partition);
    const Found f = find(c.pma.array, key, from, to);
    if (f.has && f.elem.key == key) return f.elem.val;
    return 0.0;
}

// setindex!(pcsc, value, key, partition)   src/pcsr.jl:294-339
void pcsc_set(PackedCSC& c, double value, int64_t key, int64_t partition) {
    if (partition < 1) fail(EBOUNDS, "semaphores[partition]");
    if (partition > c.semaphores.len()) {
        int64_t p = c.semaphores.len() + 1;   // _add_partitions!   :312-319
        while (p <= partition) { addpartition(c); p += 1; }
    }
    if (c.semaphores.empty_at(partition)) fail(EDELETED, "The partition has been deleted.");
    const int64_t from = c.semaphores.v[partition - 1];
    const int64_t to = pos_of_partition_end(c, partition);
    if (value != 0.0) {
        const InsRes r = insert(c.pma.array, key, value, from + 1, to, &c.semaphores);   // :321-329
        if (r.is_new) {
            c.pma.nb_elements += 1;
            const Window w = look_for_rebalance(c.pma, r.pos);
            pcsc_even_rebalance(c, w.ws, w.we, w.count);
        }
    } else {
        const InsRes r = erase(c.pma.array, key, from, to);   // :331-339
        if (r.is_new) {
            c.pma.nb_elements -= 1;
            const Window w = look_for_rebalance(c.pma, r.pos);
            pcsc_even_rebalance(c, w.ws, w.we, w.count);
        }
    }
}

// MappedPackedCSC(K, L, T)   src/pcsr.jl:82-86
void mpcsc_init_empty(MappedPackedCSC& m) {
    m = MappedPackedCSC();
    pcsc_init_empty(m.pcsc);
}

// dynamicsparsecolmajor + _dynamicsparse   src/pcsr.jl:354-449
// The reference sorts (col,row) pairs with an UNSTABLE QuickSort (:360); any
// order of equal pairs is a legal outcome — this restatement uses the stable one.
void mpcsc_init_coo(MappedPackedCSC& m, std::vector<int64_t> I, std::vector<int64_t> J,
                    std::vector<double> V, int32_t combine) {
    if (!(I.size() == J.size() && J.size() == V.size()))
        fail(EARG, "rows, columns, and nonzeros do not have same length.");
    const int64_t nnz = (int64_t)I.size();
    if (nnz == 0) { mpcsc_init_empty(m); return; }
    std::vector<int64_t> perm(nnz);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) {
        return J[x] < J[y] || (J[x] == J[y] && I[x] < I[y]);
    });
    {
        std::vector<int64_t> i2(nnz), j2(nnz); std::vector<double> v2(nnz);
        for (int64_t k = 0; k < nnz; ++k) { i2[k] = I[perm[k]]; j2[k] = J[perm[k]]; v2[k] = V[perm[k]]; }
        I.swap(i2); J.swap(j2); V.swap(v2);
    }
    // in-place duplicate combine   :365-398
    int64_t write_pos = 1, read_pos = 1;
    int64_t prev_i = I[0], prev_j = J[0];
    while (read_pos < nnz) {
        read_pos += 1;
        const int64_t cur_i = I[read_pos - 1], cur_j = J[read_pos - 1];
        if (prev_i == cur_i && prev_j == cur_j) {
            V[write_pos - 1] = combine_apply(combine, V[write_pos - 1], V[read_pos - 1]);
        } else {
            write_pos += 1;
            if (write_pos < read_pos) {
                I[write_pos - 1] = cur_i; J[write_pos - 1] = cur_j; V[write_pos - 1] = V[read_pos - 1];
            }
            prev_i = cur_i; prev_j = cur_j;
        }
    }
    I.resize(write_pos); J.resize(write_pos); V.resize(write_pos);
    // split by column   :400-421
    std::vector<int64_t> col_keys;
    std::vector<std::vector<int64_t>> row_keys;
    std::vector<std::vector<double>> values;
    for (int64_t k = 0; k < write_pos; ++k) {
        if (k == 0 || J[k] != J[k - 1]) {
            col_keys.push_back(J[k]);
            row_keys.emplace_back();
            values.emplace_back();
        }
        row_keys.back().push_back(I[k]);
        values.back().push_back(V[k]);
    }
    // MappedPackedCSC(row_keys, col_keys, values, combine)   :73-80
    m = MappedPackedCSC();
    pcsc_init(m.pcsc, row_keys, values, combine);
    for (int64_t ck : col_keys) m.col_keys.push(ck);
}

// getindex(mpcsc, row, col)   src/pcsr.jl:261-267
double mpcsc_get(const MappedPackedCSC& m, int64_t row, int64_t col) {
    const FoundKey f = find(m.col_keys, col);
    if (!(f.has && f.key == col)) return 0.0;
    return pcsc_get(m.pcsc, row, f.pos);
}

// setindex!(mpcsc, value, row, col)   src/pcsr.jl:341-351
void mpcsc_set(MappedPackedCSC& m, double value, int64_t row, int64_t col) {
    const FoundKey f = find(m.col_keys, col);
    int64_t col_pos = f.pos;
    if (!(f.has && f.key == col)) col_pos = addcolumn(m, col, f.pos);
    pcsc_set(m.pcsc, value, row, col_pos);
}

// deletecolumn!(mpcsc, col)   src/pcsr.jl:206-212
void mpcsc_deletecolumn(MappedPackedCSC& m, int64_t col) {
    const FoundKey f = find(m.col_keys, col);
    if (!(f.has && f.key == col)) fail(EARG, "column does not exist.");
    m.col_keys.live[f.pos - 1] = 0;
    deletepartition(m.pcsc, f.pos);
}

// view(mpcsc, :, col) + iterate   src/views.jl:15-35
void mpcsc_col_view(const MappedPackedCSC& m, int64_t col, std::vector<int64_t>& ks, std::vector<double>& vs) {
    ks.clear(); vs.clear();
    const FoundKey f = find(m.col_keys, col);
    if (!(f.has && f.key == col)) return;   // empty view
    const int64_t from = pos_of_partition_start(m.pcsc, f.pos) + 1;
    const int64_t to = pos_of_partition_end(m.pcsc, f.pos);
    for (int64_t pos = from; pos <= to; ++pos) {
        if (!m.pcsc.pma.array.empty_at(pos)) {
            ks.push_back(m.pcsc.pma.array.cell[pos - 1].key);
            vs.push_back(m.pcsc.pma.array.cell[pos - 1].val);
        }
    }
}

// getindex(mpcsc, row, :)   src/pcsr.jl:269-283  (elements in array order)
void mpcsc_row_slice(const MappedPackedCSC& m, int64_t row, std::vector<int64_t>& ks, std::vector<double>& vs) {
    ks.clear(); vs.clear();
    int64_t partition_id = 0;
    const Elements& a = m.pcsc.pma.array;
    for (int64_t pos = 1; pos <= a.len(); ++pos) {
        if (a.empty_at(pos)) continue;
        const Cell c = a.cell[pos - 1];
        if (c.key == SEM_KEY) partition_id = (int64_t)c.val;
        if (c.key == row) {
            ORA_ASSERT(partition_id >= 1, "element before first semaphore");
            ks.push_back(m.col_keys.v[partition_id - 1]);
            vs.push_back(c.val);
        }
    }
}

// ---------------------------------------------------------------------------
// src/buffer.jl + src/matrix.jl
// ---------------------------------------------------------------------------
// dynamicsparse(I, J, V, m, n)   src/matrix.jl:15-19  (m,n < 0 => _guess_length)
void mat_init_coo(DynMat& a, const std::vector<int64_t>& I, const std::vector<int64_t>& J,
                  const std::vector<double>& V, int64_t m, int64_t n) {
    a = DynMat();
    if (m < 0) { m = 0; for (int64_t x : I) m = std::max(m, x); }
    if (n < 0) { n = 0; for (int64_t x : J) n = std::max(n, x); }
    a.m = m; a.n = n;
    a.fillmode = false;
    mpcsc_init_coo(a.colmajor, I, J, V, COMBINE_ADD);
    mpcsc_init_coo(a.rowmajor, J, I, V, COMBINE_ADD);
    a.has_major = true;
}

// dynamicsparse(K, L, T; fill_mode)   src/matrix.jl:31-41
void mat_init_empty(DynMat& a, bool fill_mode) {
    a = DynMat();
    if (fill_mode) {
        a.fillmode = true;
        a.has_buffer = true;
    } else {
        mpcsc_init_empty(a.colmajor);
        mpcsc_init_empty(a.rowmajor);
        a.has_major = true;
    }
}

// addelem!   src/buffer.jl:20-31
static void buffer_addelem(Buffer& b, int64_t rowid, int64_t colid, double val) {
    auto it = b.index.find(rowid);
    size_t r;
    if (it == b.index.end()) {
        r = b.rowids.size();
        b.index.emplace(rowid, r);
        b.rowids.push_back(rowid);
        b.colids.emplace_back();
        b.vals.emplace_back();
    } else {
        r = it->second;
    }
    b.colids[r].push_back(colid);
    b.vals[r].push_back(val);
    b.length += 1;
}

// addrow!(buffer, ...)   src/buffer.jl:10-18
static void buffer_addrow(Buffer& b, int64_t rowid, const std::vector<int64_t>& colids, const std::vector<double>& vals) {
    if (b.index.count(rowid)) fail(EMODE, "Row already written in dynamic sparse matrix buffer.");
    const size_t n = colids.size();
    if (vals.size() != n) fail(EARG, "colids & vals must have same length.");
    std::vector<size_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](size_t x, size_t y) { return colids[x] < colids[y]; });
    const size_t r = b.rowids.size();
    b.index.emplace(rowid, r);
    b.rowids.push_back(rowid);
    b.colids.emplace_back(n);
    b.vals.emplace_back(n);
    for (size_t i = 0; i < n; ++i) { b.colids[r][i] = colids[perm[i]]; b.vals[r][i] = vals[perm[i]]; }
    b.length += (int64_t)n;
}

// setindex!(m, val, row, col)   src/matrix.jl:43-62
void mat_set(DynMat& a, double val, int64_t row, int64_t col) {
    if (val != 0.0) {
        a.m = std::max(a.m, row);
        a.n = std::max(a.n, col);
    }
    if (a.fillmode) {
        buffer_addelem(a.buffer, row, col, val);
    } else {
        mpcsc_set(a.colmajor, val, row, col);
        mpcsc_set(a.rowmajor, val, col, row);
    }
}

// getindex(m, row, col)   src/matrix.jl:64-68
double mat_get(const DynMat& a, int64_t row, int64_t col) {
    if (a.fillmode) fail(EMODE, "getindex(row, col) is not available in fill mode.");
    return mpcsc_get(a.colmajor, row, col);
}

// addrow!(matrix, ...)   src/matrix.jl:113-124
void mat_addrow(DynMat& a, int64_t row, const std::vector<int64_t>& colids, const std::vector<double>& vals) {
    if (a.fillmode) {
        buffer_addrow(a.buffer, row, colids, vals);
    } else {
        if (colids.size() != vals.size()) fail(EARG, "colids & vals must have same length.");
        for (size_t j = 0; j < colids.size(); ++j) mat_set(a, vals[j], row, colids[j]);
    }
}

// closefillmode!   src/matrix.jl:126-134  (+ get_rowids_colids_vals src/buffer.jl:33-50)
void mat_closefillmode(DynMat& a) {
    if (!a.fillmode) fail(EMODE, "Cannot close fill mode because matrix is not in fill mode.");
    std::vector<int64_t> I, J; std::vector<double> V;
    I.reserve(a.buffer.length); J.reserve(a.buffer.length); V.reserve(a.buffer.length);
    for (size_t r = 0; r < a.buffer.rowids.size(); ++r) {
        for (size_t i = 0; i < a.buffer.vals[r].size(); ++i) {
            I.push_back(a.buffer.rowids[r]);
            J.push_back(a.buffer.colids[r][i]);
            V.push_back(a.buffer.vals[r][i]);
        }
    }
    a.fillmode = false;
    a.has_buffer = false;
    a.buffer = Buffer();
    mpcsc_init_coo(a.colmajor, I, J, V, COMBINE_ADD);
    mpcsc_init_coo(a.rowmajor, J, I, V, COMBINE_ADD);
    a.has_major = true;
}

// deletecolumn!(matrix, col)   src/matrix.jl:95-102
void mat_deletecolumn(DynMat& a, int64_t col) {
    if (a.fillmode) fail(EMODE, "Cannot delete a column in fill mode");
    std::vector<int64_t> rows; std::vector<double> vals;
    mpcsc_col_view(a.colmajor, col, rows, vals);
    for (int64_t row : rows) mpcsc_set(a.rowmajor, 0.0, col, row);
    mpcsc_deletecolumn(a.colmajor, col);
}

// deleterow!(matrix, row)   src/matrix.jl:104-111
void mat_deleterow(DynMat& a, int64_t row) {
    if (a.fillmode) fail(EMODE, "Cannot delete a row in fill mode");
    std::vector<int64_t> cols; std::vector<double> vals;
    mpcsc_col_view(a.rowmajor, row, cols, vals);
    for (int64_t col : cols) mpcsc_set(a.colmajor, 0.0, row, col);
    mpcsc_deletecolumn(a.rowmajor, row);
}

// ---------------------------------------------------------------------------
// src/operations.jl
// ---------------------------------------------------------------------------
// _mul + _mul_dyn_mat_col_loop!   src/operations.jl:62-135
void mul(const MappedPackedCSC& mat, const int64_t* xi, const double* xv, int64_t nx,
         std::unordered_map<int64_t, double>& result) {
    result.clear();
    const Table& col_keys = mat.col_keys;
    const Table& sems = mat.pcsc.semaphores;
    const Elements& arr = mat.pcsc.pma.array;
    int64_t col_key_pos = 1;
    for (int64_t e = 0; e < nx; ++e) {
        const int64_t vec_row_id = xi[e];
        const double vec_val = xv[e];
        // :64-70 advance to the first live column key >= vec_row_id
        while (col_key_pos <= col_keys.len()) {
            if (!col_keys.empty_at(col_key_pos) && col_keys.v[col_key_pos - 1] >= vec_row_id) break;
            col_key_pos += 1;
        }
        if (col_key_pos > col_keys.len()) break;   // :72-74
        if (col_keys.empty_at(col_key_pos) || col_keys.v[col_key_pos - 1] != vec_row_id) continue;   // :76-79
        int64_t next_col_key_pos = col_key_pos + 1;   // :81-84
        while (next_col_key_pos <= col_keys.len() && col_keys.empty_at(next_col_key_pos)) next_col_key_pos += 1;
        ORA_ASSERT(!sems.empty_at(col_key_pos), "!isnothing(cur_semaphore)");
        const int64_t mat_row_start = sems.v[col_key_pos - 1] + 1;
        int64_t mat_row_end = arr.len();
        if (next_col_key_pos <= col_keys.len()) {
            ORA_ASSERT(!sems.empty_at(next_col_key_pos), "!isnothing(next_semaphore)");
            mat_row_end = sems.v[next_col_key_pos - 1] - 1;
        }
        for (int64_t pos = mat_row_start; pos <= mat_row_end; ++pos) {   // :97-103
            if (!arr.empty_at(pos)) {
                const Cell c = arr.cell[pos - 1];
                auto it = result.find(c.key);
                const double prev = (it == result.end()) ? 0.0 : it->second;
                const double prod = vec_val * c.val;
                result[c.key] = prev + prod;
            }
        }
        col_key_pos = next_col_key_pos;
    }
}

// ---------------------------------------------------------------------------
// digests
// ---------------------------------------------------------------------------
uint64_t layout_digest(const Elements& a) {
    uint64_t h = 1469598103934665603ULL;
    for (int64_t pos = 1; pos <= a.len(); ++pos) {
        if (a.empty_at(pos)) continue;
        h = (h ^ (uint64_t)pos) * 1099511628211ULL;
        h = (h ^ (uint64_t)a.cell[pos - 1].key) * 1099511628211ULL;
    }
    return h;
}
uint64_t table_digest(const Table& t) {
    uint64_t h = 1469598103934665603ULL;
    for (int64_t i = 1; i <= t.len(); ++i) {
        const uint64_t v = t.empty_at(i) ? 0ULL : (uint64_t)t.v[i - 1];
        h = (h ^ v) * 1099511628211ULL;
    }
    return h;
}

}  // namespace ora
