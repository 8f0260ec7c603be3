// oracle/oracle_capi.cpp — CPU ORACLE C API.  TEST INFRASTRUCTURE ONLY (see oracle.hpp).
//
// Exposes the restatement through the same call shapes as include/dsa.h, with the
// prefix `ora_` instead of `dsa_`, so a test can drive the HIP library and the
// oracle with identical scripts and compare layouts slot for slot.  Extra `ora_raw_*`
// entry points expose the slot-array primitives for the reference's unit-test vectors.
#include "oracle.hpp"

#include <algorithm>
#include <cstring>

using namespace ora;

static thread_local std::string g_err;

#define ORA_TRY try {
#define ORA_CATCH                                                         \
    } catch (const Err& e) { g_err = e.msg; return e.code;               \
    } catch (const std::exception& e) { g_err = e.what(); return EASSERT; } \
    return OK;

struct ora_vec { DynVec v; };
struct ora_pcsc { PackedCSC c; };
struct ora_mat { DynMat a; };

static void fill_info(const PMA& p, int64_t nb_partitions, int64_t table_len, int64_t* info) {
    std::memset(info, 0, sizeof(int64_t) * 16);
    info[0] = p.capacity; info[1] = p.segment_capacity; info[2] = p.nb_segments;
    info[3] = p.nb_elements; info[4] = p.height; info[5] = nb_partitions; info[6] = table_len;
    info[7] = p.stat_window_slots; info[8] = p.stat_rebalances; info[9] = p.stat_extends;
    info[10] = p.stat_shrinks;
}

static void export_elements(const Elements& a, int64_t* keys, double* vals, uint8_t* occ, int64_t cap) {
    if (cap < a.len()) throw Err{ECAP, "output buffers smaller than capacity"};
    for (int64_t i = 0; i < a.len(); ++i) {
        occ[i] = a.tag[i];
        keys[i] = a.tag[i] ? a.cell[i].key : 0;
        vals[i] = a.tag[i] ? a.cell[i].val : 0.0;
    }
}

extern "C" {

const char* ora_last_error_message(void) { return g_err.c_str(); }

// ---------------- raw slot-array primitives (unit-test vectors) ----------------
static Elements raw_load(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t len) {
    Elements a; a.resize(len);
    for (int64_t i = 0; i < len; ++i) if (occ[i]) a.set(i + 1, keys[i], vals[i]);
    return a;
}
static void raw_store(const Elements& a, int64_t* keys, double* vals, uint8_t* occ) {
    for (int64_t i = 0; i < a.len(); ++i) {
        occ[i] = a.tag[i];
        keys[i] = a.tag[i] ? a.cell[i].key : 0;
        vals[i] = a.tag[i] ? a.cell[i].val : 0.0;
    }
}

int32_t ora_raw_find(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t len,
                     int64_t key, int64_t from, int64_t to, int64_t* pos, int32_t* has,
                     int64_t* fkey, double* fval) {
    ORA_TRY
    const Elements a = raw_load(keys, vals, occ, len);
    const Found f = find(a, key, from, to);
    *pos = f.pos; *has = f.has ? 1 : 0; *fkey = f.elem.key; *fval = f.elem.val;
    ORA_CATCH
}

int32_t ora_raw_insert(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t key, double value,
                       int64_t from, int64_t to, int64_t* sems, int64_t nsems, int64_t* pos, int32_t* is_new) {
    ORA_TRY
    Elements a = raw_load(keys, vals, occ, len);
    Table t; if (sems) { t.resize(nsems); for (int64_t i = 0; i < nsems; ++i) { t.v[i] = sems[i]; t.live[i] = sems[i] != 0; } }
    const InsRes r = insert(a, key, value, from, to, sems ? &t : nullptr);
    *pos = r.pos; *is_new = r.is_new ? 1 : 0;
    raw_store(a, keys, vals, occ);
    if (sems) for (int64_t i = 0; i < nsems; ++i) sems[i] = t.v[i];
    ORA_CATCH
}

int32_t ora_raw_delete(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t key,
                       int64_t from, int64_t to, int64_t* pos, int32_t* deleted) {
    ORA_TRY
    Elements a = raw_load(keys, vals, occ, len);
    const InsRes r = erase(a, key, from, to);
    *pos = r.pos; *deleted = r.is_new ? 1 : 0;
    raw_store(a, keys, vals, occ);
    ORA_CATCH
}

int32_t ora_raw_purge(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t from, int64_t to,
                      int64_t* mid, int64_t* nb) {
    ORA_TRY
    Elements a = raw_load(keys, vals, occ, len);
    const PurgeRes r = purge(a, from, to);
    *mid = r.mid; *nb = r.nb;
    raw_store(a, keys, vals, occ);
    ORA_CATCH
}

int32_t ora_raw_move(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int32_t to_right,
                     int64_t from, int64_t to, int64_t* sems, int64_t nsems) {
    ORA_TRY
    Elements a = raw_load(keys, vals, occ, len);
    Table t; if (sems) { t.resize(nsems); for (int64_t i = 0; i < nsems; ++i) { t.v[i] = sems[i]; t.live[i] = sems[i] != 0; } }
    if (to_right) movecellstoright(a, from, to, sems ? &t : nullptr);
    else movecellstoleft(a, from, to, sems ? &t : nullptr);
    raw_store(a, keys, vals, occ);
    if (sems) for (int64_t i = 0; i < nsems; ++i) sems[i] = t.v[i];
    ORA_CATCH
}

// pack! then spread! of window [ws, we] holding m cells (src/moves.jl:94-171)
int32_t ora_raw_pack_spread(int64_t* keys, double* vals, uint8_t* occ, int64_t len, int64_t ws, int64_t we,
                            int64_t m, int64_t* sems, int64_t nsems, int32_t do_pack, int32_t do_spread) {
    ORA_TRY
    Elements a = raw_load(keys, vals, occ, len);
    Table t; if (sems) { t.resize(nsems); for (int64_t i = 0; i < nsems; ++i) { t.v[i] = sems[i]; t.live[i] = sems[i] != 0; } }
    if (do_pack) pack(a, ws, we, m);
    if (do_spread) { if (sems) spread_sem(a, ws, we, m, &t); else spread(a, ws, we, m); }
    raw_store(a, keys, vals, occ);
    if (sems) for (int64_t i = 0; i < nsems; ++i) sems[i] = t.v[i];
    ORA_CATCH
}

uint64_t ora_raw_layout_digest(const int64_t* keys, const uint8_t* occ, int64_t len) {
    uint64_t h = 1469598103934665603ULL;
    for (int64_t pos = 1; pos <= len; ++pos) {
        if (!occ[pos - 1]) continue;
        h = (h ^ (uint64_t)pos) * 1099511628211ULL;
        h = (h ^ (uint64_t)keys[pos - 1]) * 1099511628211ULL;
    }
    return h;
}
uint64_t ora_raw_table_digest(const int64_t* t, int64_t len) {
    uint64_t h = 1469598103934665603ULL;
    for (int64_t i = 0; i < len; ++i) h = (h ^ (uint64_t)t[i]) * 1099511628211ULL;
    return h;
}

// ---------------- vector ----------------
int32_t ora_vec_create(const int64_t* keys, const double* vals, int64_t n, int32_t combine_op,
                       int64_t len, ora_vec** out) {
    ORA_TRY
    auto* h = new ora_vec();
    try {
        vec_init(h->v, std::vector<int64_t>(keys, keys + n), std::vector<double>(vals, vals + n), combine_op, len);
    } catch (...) { delete h; throw; }
    *out = h;
    ORA_CATCH
}
int32_t ora_vec_create_empty(ora_vec** out) {
    ORA_TRY
    auto* h = new ora_vec();
    vec_init(h->v, {}, {}, COMBINE_ADD, -1);
    *out = h;
    ORA_CATCH
}
int32_t ora_vec_destroy(ora_vec* h) { delete h; return OK; }
int32_t ora_vec_get(ora_vec* h, int64_t key, double* out) { ORA_TRY *out = pma_get(h->v.pma, key); ORA_CATCH }
int32_t ora_vec_get_batch(ora_vec* h, const int64_t* keys, int64_t n, double* out) {
    ORA_TRY for (int64_t i = 0; i < n; ++i) out[i] = pma_get(h->v.pma, keys[i]); ORA_CATCH
}
int32_t ora_vec_set(ora_vec* h, int64_t key, double val) { ORA_TRY vec_set(h->v, key, val); ORA_CATCH }
int32_t ora_vec_set_batch(ora_vec* h, const int64_t* keys, const double* vals, int64_t n) {
    ORA_TRY for (int64_t i = 0; i < n; ++i) vec_set(h->v, keys[i], vals[i]); ORA_CATCH
}
int32_t ora_vec_nnz(ora_vec* h, int64_t* out) { *out = h->v.pma.nb_elements; return OK; }
int32_t ora_vec_len(ora_vec* h, int64_t* out) { *out = h->v.n; return OK; }
int32_t ora_vec_shrink_size(ora_vec* h) {   // shrink_size!  src/vector.jl:64 (+ _guess_length :7-8)
    int64_t n = 0;
    const Elements& a = h->v.pma.array;
    for (int64_t i = 0; i < a.len(); ++i) if (a.tag[i]) n = std::max(n, a.cell[i].key);
    h->v.n = n;
    return OK;
}
int32_t ora_vec_nonzeros(ora_vec* h, int64_t* keys, double* vals, int64_t cap, int64_t* n_out) {
    ORA_TRY
    const Elements& a = h->v.pma.array;
    int64_t n = 0;
    for (int64_t i = 0; i < a.len(); ++i) if (a.tag[i]) {
        if (n >= cap) throw Err{ECAP, "output buffers too small"};
        keys[n] = a.cell[i].key; vals[n] = a.cell[i].val; ++n;
    }
    *n_out = n;
    ORA_CATCH
}
// _arrays_equal(array1, array2)  src/pma.jl:236-260: both slot arrays are walked skipping `nothing`; the i-th stored tuples
// must be == (Int key ==, Float64 value ==), and neither side may have a stored tuple left over.
static bool arrays_equal(const Elements& a1, const Elements& a2) {
    int64_t i = 0, j = 0;
    const int64_t n1 = a1.len(), n2 = a2.len();
    while (true) {
        while (i < n1 && !a1.tag[i]) ++i;
        while (j < n2 && !a2.tag[j]) ++j;
        if (i >= n1 || j >= n2) return i >= n1 && j >= n2;
        if (a1.cell[i].key != a2.cell[j].key || !(a1.cell[i].val == a2.cell[j].val)) return false;
        ++i; ++j;
    }
}
// _arrays_equal on two raw slot arrays (golden vectors of test/unit/comparison.jl:2-10)
int32_t ora_raw_arrays_equal(const int64_t* k1, const double* v1, const uint8_t* o1, int64_t n1, const int64_t* k2, const double* v2,
                             const uint8_t* o2, int64_t n2, int32_t* out) {
    ORA_TRY
    const Elements a1 = raw_load(k1, v1, o1, n1), a2 = raw_load(k2, v2, o2, n2);
    *out = arrays_equal(a1, a2) ? 1 : 0;
    ORA_CATCH
}
// v1 == v2  src/vector.jl:85-87 ; pma1 == pma2  src/pma.jl:262-266
int32_t ora_vec_equal(ora_vec* a, ora_vec* b, int32_t* out) {
    ORA_TRY
    *out = 0;
    if (a->v.n != b->v.n) return OK;
    if (a == b) { *out = 1; return OK; }
    if (a->v.pma.nb_elements != b->v.pma.nb_elements) return OK;
    *out = arrays_equal(a->v.pma.array, b->v.pma.array) ? 1 : 0;
    ORA_CATCH
}
// alpha*a + beta*b as ascending (key, value) pairs: what v1 + v2, v1 - v2 and -v evaluate to in the reference through the
// AbstractSparseVector fallbacks over nonzeroinds / nonzeros (src/vector.jl:93-109; exercised by test/functional/math.jl:53-94).
// The merge itself is SparseArrays stdlib code (not under /root/reference): union of the stored keys, a key stored in both
// operands is dropped when the combined value is zero; restated here as a two-pointer merge.
int32_t ora_vec_axpby(ora_vec* a, double alpha, ora_vec* b, double beta, int64_t* keys, double* vals, int64_t cap, int64_t* n_out) {
    ORA_TRY
    std::vector<Cell> x, y;
    { const Elements& e = a->v.pma.array; for (int64_t i = 0; i < e.len(); ++i) if (e.tag[i]) x.push_back(e.cell[i]); }
    { const Elements& e = b->v.pma.array; for (int64_t i = 0; i < e.len(); ++i) if (e.tag[i]) y.push_back(e.cell[i]); }
    size_t i = 0, j = 0; int64_t n = 0;
    auto put = [&](int64_t k, double v) { if (n >= cap) throw Err{ECAP, "output buffers too small"}; keys[n] = k; vals[n] = v; ++n; };
    while (i < x.size() || j < y.size()) {
        if (j >= y.size() || (i < x.size() && x[i].key < y[j].key)) { put(x[i].key, alpha * x[i].val); ++i; }
        else if (i >= x.size() || y[j].key < x[i].key) { put(y[j].key, beta * y[j].val); ++j; }
        else { const double v = alpha * x[i].val + beta * y[j].val; if (v != 0.0) put(x[i].key, v); ++i; ++j; }
    }
    *n_out = n;
    ORA_CATCH
}
int32_t ora_vec_info(ora_vec* h, int64_t* info) { fill_info(h->v.pma, h->v.n, 0, info); return OK; }
int32_t ora_vec_export_layout(ora_vec* h, int64_t* keys, double* vals, uint8_t* occ, int64_t cap) {
    ORA_TRY export_elements(h->v.pma.array, keys, vals, occ, cap); ORA_CATCH
}
int32_t ora_vec_rebalance_root(ora_vec* h) {
    ORA_TRY even_rebalance(h->v.pma, 1, h->v.pma.capacity, h->v.pma.nb_elements); ORA_CATCH
}

// ---------------- PackedCSC ----------------
int32_t ora_pcsc_create(const int64_t* colptr, int64_t nparts, const int64_t* row_keys, const double* vals,
                        int32_t combine_op, ora_pcsc** out) {
    ORA_TRY
    if (nparts <= 0) throw Err{EARG, "PackedCSC needs at least one partition"};
    std::vector<std::vector<int64_t>> rk(nparts); std::vector<std::vector<double>> vv(nparts);
    for (int64_t p = 0; p < nparts; ++p) {
        rk[p].assign(row_keys + colptr[p], row_keys + colptr[p + 1]);
        vv[p].assign(vals + colptr[p], vals + colptr[p + 1]);
    }
    auto* h = new ora_pcsc();
    try { pcsc_init(h->c, rk, vv, combine_op); } catch (...) { delete h; throw; }
    *out = h;
    ORA_CATCH
}
int32_t ora_pcsc_create_empty(ora_pcsc** out) { ORA_TRY auto* h = new ora_pcsc(); pcsc_init_empty(h->c); *out = h; ORA_CATCH }
int32_t ora_pcsc_destroy(ora_pcsc* h) { delete h; return OK; }
int32_t ora_pcsc_get(ora_pcsc* h, int64_t key, int64_t partition, double* out) { ORA_TRY *out = pcsc_get(h->c, key, partition); ORA_CATCH }
int32_t ora_pcsc_set(ora_pcsc* h, double val, int64_t key, int64_t partition) { ORA_TRY pcsc_set(h->c, val, key, partition); ORA_CATCH }
int32_t ora_pcsc_deletepartition(ora_pcsc* h, int64_t partition) { ORA_TRY deletepartition(h->c, partition); ORA_CATCH }
int32_t ora_pcsc_nnz(ora_pcsc* h, int64_t* out) { *out = h->c.pma.nb_elements - h->c.nb_partitions; return OK; }
int32_t ora_pcsc_nbpartitions(ora_pcsc* h, int64_t* out) { *out = h->c.nb_partitions; return OK; }
int32_t ora_pcsc_info(ora_pcsc* h, int64_t* info) { fill_info(h->c.pma, h->c.nb_partitions, h->c.semaphores.len(), info); return OK; }
int32_t ora_pcsc_export_layout(ora_pcsc* h, int64_t* keys, double* vals, uint8_t* occ, int64_t cap,
                               int64_t* semaphores, int64_t table_cap) {
    ORA_TRY
    export_elements(h->c.pma.array, keys, vals, occ, cap);
    if (table_cap < h->c.semaphores.len()) throw Err{ECAP, "semaphore buffer too small"};
    for (int64_t i = 0; i < h->c.semaphores.len(); ++i) semaphores[i] = h->c.semaphores.live[i] ? h->c.semaphores.v[i] : 0;
    ORA_CATCH
}

// ---------------- matrix ----------------
static void check_key(int64_t k) { if (k == 0) throw Err{EKEY, "0 is the reserved semaphore key"}; }
static MappedPackedCSC& orient(ora_mat* h, int32_t o) {
    if (!h->a.has_major) throw Err{EMODE, "matrix is in fill mode"};
    return o == 0 ? h->a.colmajor : h->a.rowmajor;
}

int32_t ora_mat_create_from_coo(const int64_t* I, const int64_t* J, const double* V, int64_t nnz,
                                int64_t m, int64_t n, ora_mat** out) {
    ORA_TRY
    for (int64_t k = 0; k < nnz; ++k) { check_key(I[k]); check_key(J[k]); }
    auto* h = new ora_mat();
    try {
        mat_init_coo(h->a, std::vector<int64_t>(I, I + nnz), std::vector<int64_t>(J, J + nnz),
                     std::vector<double>(V, V + nnz), m, n);
    } catch (...) { delete h; throw; }
    *out = h;
    ORA_CATCH
}
int32_t ora_mat_create_empty(int32_t fill_mode, ora_mat** out) {
    ORA_TRY auto* h = new ora_mat(); mat_init_empty(h->a, fill_mode != 0); *out = h; ORA_CATCH
}
int32_t ora_mat_destroy(ora_mat* h) { delete h; return OK; }
int32_t ora_mat_set(ora_mat* h, double val, int64_t row, int64_t col) {
    ORA_TRY check_key(row); check_key(col); mat_set(h->a, val, row, col); ORA_CATCH
}
int32_t ora_mat_set_batch(ora_mat* h, const int64_t* I, const int64_t* J, const double* V, int64_t n) {
    ORA_TRY for (int64_t k = 0; k < n; ++k) { check_key(I[k]); check_key(J[k]); mat_set(h->a, V[k], I[k], J[k]); } ORA_CATCH
}
int32_t ora_mat_get(ora_mat* h, int64_t row, int64_t col, double* out) { ORA_TRY *out = mat_get(h->a, row, col); ORA_CATCH }
int32_t ora_mat_get_batch(ora_mat* h, const int64_t* I, const int64_t* J, int64_t n, double* out) {
    ORA_TRY for (int64_t k = 0; k < n; ++k) out[k] = mat_get(h->a, I[k], J[k]); ORA_CATCH
}
int32_t ora_mat_addrow(ora_mat* h, int64_t row, const int64_t* colids, const double* vals, int64_t n) {
    ORA_TRY
    check_key(row); for (int64_t k = 0; k < n; ++k) check_key(colids[k]);
    mat_addrow(h->a, row, std::vector<int64_t>(colids, colids + n), std::vector<double>(vals, vals + n));
    ORA_CATCH
}
int32_t ora_mat_closefillmode(ora_mat* h) { ORA_TRY mat_closefillmode(h->a); ORA_CATCH }
int32_t ora_mat_deletecolumn(ora_mat* h, int64_t col) { ORA_TRY mat_deletecolumn(h->a, col); ORA_CATCH }
int32_t ora_mat_deleterow(ora_mat* h, int64_t row) { ORA_TRY mat_deleterow(h->a, row); ORA_CATCH }
static int32_t view_impl(ora_mat* h, int32_t o, int64_t key, int64_t* ks, double* vs, int64_t cap, int64_t* n_out) {
    ORA_TRY
    if (h->a.fillmode) throw Err{EMODE, "View not available in fill mode."};
    std::vector<int64_t> k; std::vector<double> v;
    mpcsc_col_view(orient(h, o), key, k, v);
    if ((int64_t)k.size() > cap) throw Err{ECAP, "output buffers too small"};
    std::copy(k.begin(), k.end(), ks); std::copy(v.begin(), v.end(), vs);
    *n_out = (int64_t)k.size();
    ORA_CATCH
}
int32_t ora_mat_col_view(ora_mat* h, int64_t col, int64_t* rows, double* vals, int64_t cap, int64_t* n_out) {
    return view_impl(h, 0, col, rows, vals, cap, n_out);
}
int32_t ora_mat_row_view(ora_mat* h, int64_t row, int64_t* cols, double* vals, int64_t cap, int64_t* n_out) {
    return view_impl(h, 1, row, cols, vals, cap, n_out);
}
// m[:, col] / m[row, :] as new vectors.  The row slice follows the reference literally: a scan of the COLMAJOR array
// (mpcsc_row_slice, src/pcsr.jl:269-283) — the HIP library serves it from the rowmajor twin; both must agree.
int32_t ora_mat_col_slice(ora_mat* h, int64_t col, ora_vec** out) {
    ORA_TRY
    if (h->a.fillmode) throw Err{EMODE, "slices are not available in fill mode"};
    std::vector<int64_t> k; std::vector<double> v;
    mpcsc_col_view(orient(h, 0), col, k, v);
    auto* r = new ora_vec();
    vec_init(r->v, k, v, COMBINE_ADD, -1);
    *out = r;
    ORA_CATCH
}
int32_t ora_mat_row_slice(ora_mat* h, int64_t row, ora_vec** out) {
    ORA_TRY
    if (h->a.fillmode) throw Err{EMODE, "slices are not available in fill mode"};
    std::vector<int64_t> k; std::vector<double> v;
    mpcsc_row_slice(orient(h, 0), row, k, v);
    auto* r = new ora_vec();
    vec_init(r->v, k, v, COMBINE_ADD, -1);
    *out = r;
    ORA_CATCH
}
int32_t ora_mat_nnz(ora_mat* h, int64_t* out) {   // nnz(m) = nnz(m.rowmajor)  src/matrix.jl:91
    ORA_TRY
    const MappedPackedCSC& r = orient(h, 1);
    *out = r.pcsc.pma.nb_elements - r.pcsc.nb_partitions;
    ORA_CATCH
}
int32_t ora_mat_size(ora_mat* h, int64_t* m, int64_t* n) { *m = h->a.m; *n = h->a.n; return OK; }
int32_t ora_mat_nbpartitions(ora_mat* h, int32_t o, int64_t* out) { ORA_TRY *out = orient(h, o).pcsc.nb_partitions; ORA_CATCH }
int32_t ora_mat_info(ora_mat* h, int32_t o, int64_t* info) {
    ORA_TRY const MappedPackedCSC& c = orient(h, o); fill_info(c.pcsc.pma, c.pcsc.nb_partitions, c.pcsc.semaphores.len(), info); ORA_CATCH
}
int32_t ora_mat_export_layout(ora_mat* h, int32_t o, int64_t* keys, double* vals, uint8_t* occ, int64_t cap,
                              int64_t* semaphores, int64_t* col_keys, uint8_t* col_live, int64_t table_cap) {
    ORA_TRY
    const MappedPackedCSC& c = orient(h, o);
    export_elements(c.pcsc.pma.array, keys, vals, occ, cap);
    const int64_t tl = c.pcsc.semaphores.len();
    if (table_cap < tl || table_cap < c.col_keys.len()) throw Err{ECAP, "table buffers too small"};
    for (int64_t i = 0; i < tl; ++i) semaphores[i] = c.pcsc.semaphores.live[i] ? c.pcsc.semaphores.v[i] : 0;
    for (int64_t i = 0; i < c.col_keys.len(); ++i) { col_keys[i] = c.col_keys.live[i] ? c.col_keys.v[i] : 0; col_live[i] = c.col_keys.live[i]; }
    ORA_CATCH
}
int32_t ora_mat_rebalance_root(ora_mat* h, int32_t o) {
    ORA_TRY MappedPackedCSC& c = orient(h, o); pcsc_even_rebalance(c.pcsc, 1, c.pcsc.pma.capacity, c.pcsc.pma.nb_elements); ORA_CATCH
}

// mat * v / transpose(mat) * v through the reference's Dict accumulator
// (src/operations.jl:14-36: colmajor for mat*v, rowmajor for transpose(mat)*v)
int32_t ora_mat_spmv_sparse(ora_mat* h, int32_t transpose, const int64_t* xi, const double* xv, int64_t nx,
                            int64_t* yi, double* yv, int64_t cap, int64_t* n_out) {
    ORA_TRY
    std::unordered_map<int64_t, double> result;
    mul(orient(h, transpose ? 1 : 0), xi, xv, nx, result);
    std::vector<std::pair<int64_t, double>> out(result.begin(), result.end());
    std::sort(out.begin(), out.end());   // sparsevec(result, n)  src/operations.jl:12
    if ((int64_t)out.size() > cap) throw Err{ECAP, "output buffers too small"};
    const int64_t dim = transpose ? h->a.n : h->a.m;
    for (size_t i = 0; i < out.size(); ++i) {
        if (out[i].first < 1 || out[i].first > dim) throw Err{EBOUNDS, "result index outside 1:size (sparsevec)"};
        yi[i] = out[i].first; yv[i] = out[i].second;
    }
    *n_out = (int64_t)out.size();
    ORA_CATCH
}
int32_t ora_mat_spmv_dense(ora_mat* h, int32_t transpose, const double* x, int64_t nx, double* y, int64_t ny) {
    ORA_TRY
    std::vector<int64_t> xi(nx);
    for (int64_t i = 0; i < nx; ++i) xi[i] = i + 1;
    std::unordered_map<int64_t, double> result;
    mul(orient(h, transpose ? 1 : 0), xi.data(), x, nx, result);
    std::fill(y, y + ny, 0.0);
    for (const auto& kv : result) if (kv.first >= 1 && kv.first <= ny) y[kv.first - 1] = kv.second;
    ORA_CATCH
}
// CPU-baseline variant: the same loop nest with a dense accumulator instead of the Dict
// (what a tuned single-thread CPU code would do; reported next to the Dict figure).
int32_t ora_mat_spmv_dense_fastacc(ora_mat* h, int32_t transpose, const double* x, int64_t nx, double* y, int64_t ny) {
    ORA_TRY
    const MappedPackedCSC& mat = orient(h, transpose ? 1 : 0);
    std::fill(y, y + ny, 0.0);
    const Table& ck = mat.col_keys; const Table& sems = mat.pcsc.semaphores; const Elements& arr = mat.pcsc.pma.array;
    for (int64_t p = 1; p <= ck.len(); ++p) {
        if (ck.empty_at(p)) continue;
        const int64_t col = ck.v[p - 1];
        if (col < 1 || col > nx) continue;
        const double xv = x[col - 1];
        const int64_t from = sems.v[p - 1] + 1;
        const int64_t to = pos_of_partition_end(mat.pcsc, p);
        for (int64_t pos = from; pos <= to; ++pos) {
            if (!arr.empty_at(pos)) {
                const Cell c = arr.cell[pos - 1];
                if (c.key >= 1 && c.key <= ny) { const double prod = xv * c.val; y[c.key - 1] = y[c.key - 1] + prod; }
            }
        }
    }
    ORA_CATCH
}


// ---------------- layouts restored from an export (tests put both implementations into the same arbitrary state) ----------------
// geometry from capacity and segment capacity as in src/pma.jl:42-49,143-161
static void import_pma(PMA& p, const int64_t* keys, const double* vals, const uint8_t* occ, int64_t capacity, int64_t segment_capacity) {
    auto pow2 = [](int64_t x) { return x > 0 && (x & (x - 1)) == 0; };
    if (!pow2(capacity) || !pow2(segment_capacity) || segment_capacity > capacity / 2)
        throw Err{EARG, "capacity and segment capacity must be powers of two with at least two segments"};
    p = PMA();
    p.capacity = capacity; p.segment_capacity = segment_capacity; p.nb_segments = capacity / segment_capacity;
    p.height = 0; while (((int64_t)1 << p.height) < p.nb_segments) ++p.height;
    p.t_d = (p.t_h - p.t_0) / (double)p.height;
    p.p_d = (p.p_h - p.p_0) / (double)p.height;
    p.array.resize(capacity);
    for (int64_t i = 0; i < capacity; ++i) if (occ[i]) { p.array.set(i + 1, keys[i], vals[i]); p.nb_elements += 1; }
}
int32_t ora_vec_import_layout(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t capacity, int64_t segment_capacity,
                              int64_t len, ora_vec** out) {
    ORA_TRY
    auto* h = new ora_vec();
    try { import_pma(h->v.pma, keys, vals, occ, capacity, segment_capacity); } catch (...) { delete h; throw; }
    h->v.n = len;
    *out = h;
    ORA_CATCH
}
int32_t ora_pcsc_import_layout(const int64_t* keys, const double* vals, const uint8_t* occ, int64_t capacity, int64_t segment_capacity,
                               const int64_t* semaphores, int64_t table_len, ora_pcsc** out) {
    ORA_TRY
    auto* h = new ora_pcsc();
    try {
        import_pma(h->c.pma, keys, vals, occ, capacity, segment_capacity);
        h->c.semaphores.resize(table_len);
        for (int64_t i = 0; i < table_len; ++i) {
            h->c.semaphores.v[i] = semaphores[i]; h->c.semaphores.live[i] = semaphores[i] != 0;
            if (semaphores[i] != 0) h->c.nb_partitions += 1;
        }
    } catch (...) { delete h; throw; }
    *out = h;
    ORA_CATCH
}

}  // extern "C"
