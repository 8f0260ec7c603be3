"""CPU property tests of the oracle with our own RNG — the reference's functional tests that use its
MersenneTwister stream (not reproducible outside Julia) re-run as value-parity checks against dict /
scipy oracles plus the structural invariants of test/utils.jl:68-113."""
import numpy as np
import pytest
import scipy.sparse as sp

from util import SplitMix64, check_key_order, check_semaphores


@pytest.mark.parametrize("n", [20, 100, 1000, 10000])
def test_vec_fill_then_empty_tracks_nnz(dsa, oracle, n):
    """test/functional/sparsevector.jl:88-119 (dynsparsevec_fill_empty)."""
    g = SplitMix64(n)
    kv = {}
    while len(kv) < n:
        kv[1 + g.next() % 10 ** 10] = g.unit12()
    v = dsa.dynamicsparsevec([], [], binding=oracle)
    for i, (k, x) in enumerate(kv.items(), start=1):
        v[k] = x
        assert v.nnz() == i
    assert all(v[k] == x for k, x in list(kv.items())[:200])
    ks, vs = v.nonzeros()
    assert np.all(np.diff(ks) > 0) and len(ks) == n
    for i, k in enumerate(kv, start=1):
        v[k] = 0.0
        assert v.nnz() == n - i
    assert v.info()["capacity"] >= 2


def test_vec_build_then_million_style_inserts(dsa, oracle):
    """test/functional/sparsevector.jl:121-161 scaled: bulk build, overwrite/insert merge, dense 1..N fill."""
    g = SplitMix64(99)
    kv1 = {1 + g.next() % 10 ** 10: g.unit12() for _ in range(20000)}
    v = dsa.dynamicsparsevec(list(kv1), list(kv1.values()), binding=oracle)
    kv2 = {1 + g.next() % 10 ** 10: g.unit12() for _ in range(20000)}
    v.set_batch(list(kv2), list(kv2.values()))
    kv1.update(kv2)
    q = list(kv1)[:: 7]
    assert np.array_equal(v.get_batch(q), np.array([kv1[k] for k in q]))
    w = dsa.dynamicsparsevec([5, 77, 4000], [1.0, 2.0, 3.0], binding=oracle)
    w.set_batch(np.arange(1, 5001), np.full(5000, 10.0))
    assert np.all(w.get_batch(np.arange(1, 5001)) == 10.0) and w.nnz() == 5000


def test_pcsc_42_partitions_random_accumulate(dsa, oracle):
    """test/functional/sparsematrix.jl:123-156 (pcsc_insertions_and_gets)."""
    g = SplitMix64(42)
    parts = [{1 + g.next() % 10000: float(1 + g.next() % 99) for _ in range(20 + g.next() % 300)} for _ in range(42)]
    p = dsa.packedcsc([list(d) for d in parts], [list(d.values()) for d in parts], binding=oracle)
    for _ in range(3000):
        pid = 1 + g.next() % 42
        key = 1 + g.next() % 10000
        assert p[key, pid] == parts[pid - 1].get(key, 0.0)
    for _ in range(5000):
        pid = 1 + g.next() % 42
        key = 1 + g.next() % 10000
        val = float(1 + g.next() % 99)
        p[key, pid] = p[key, pid] + val
        parts[pid - 1][key] = parts[pid - 1].get(key, 0.0) + val
    for pid, d in enumerate(parts, start=1):
        for key, val in list(d.items())[:100]:
            assert p[key, pid] == val
    k, v, o, s = p.export_layout()
    assert check_semaphores(k, v, o, s) == 42
    check_key_order(k, o)


def test_matrix_value_parity_with_scipy_and_new_columns(dsa, oracle):
    """test/functional/sparsematrix.jl:363-382: build, read back, then append thousands of new columns."""
    g = SplitMix64(5)
    nr, nc = 340, 1000
    I, J, V = [], [], []
    seen = set()
    for _ in range(17000):
        i, j = 1 + g.next() % nr, 1 + g.next() % nc
        if (i, j) not in seen:
            seen.add((i, j)); I.append(i); J.append(j); V.append(float(g.next() % 10 ** 6) / 1000.0)
    a = dsa.dynamicsparse(I, J, V, binding=oracle)
    assert np.array_equal(a.get_batch(I, J), np.array(V))
    cols = np.arange(nc, 4001)
    a.set_batch(np.ones(len(cols), dtype=np.int64), cols, np.ones(len(cols)))
    assert np.all(a.get_batch(np.ones(len(cols), dtype=np.int64), cols) == 1.0)
    for o in (0, 1):
        L = a.export_layout(o)
        check_semaphores(L["keys"], L["vals"], L["occ"], L["semaphores"])
        check_key_order(L["keys"], L["occ"])
    A = sp.csr_matrix((V, (np.array(I) - 1, np.array(J) - 1)), shape=(nr, 4001)).tolil()
    A[0, nc - 1:4000] = 1.0
    x = 1.0 + np.arange(4001) / 4001.0
    np.testing.assert_allclose(a.mul(x), A.tocsr() @ x, rtol=1e-12)


def test_fill_mode_random_flush_sums_duplicates(dsa, oracle):
    """test/functional/sparsematrix.jl:469-488: 10k random writes in fill mode accumulate like sparse(I,J,V)."""
    g = SplitMix64(8)
    row = [1 + g.next() % 1000 for _ in range(10000)]
    col = [1 + g.next() % 1000 for _ in range(10000)]
    val = [float(1 + g.next() % 100000) for _ in range(10000)]
    a = dsa.dynamicsparse(fill_mode=True, binding=oracle)
    a.set_batch(row, col, val)
    a.closefillmode()
    ref = sp.coo_matrix((val, (np.array(row) - 1, np.array(col) - 1)), shape=(1000, 1000)).tocsr()
    qi = np.repeat(np.arange(1, 101), 1000)
    qj = np.tile(np.arange(1, 1001), 100)
    assert np.array_equal(a.get_batch(qi, qj), np.asarray(ref[:100].todense()).ravel())
    # non-fill mode overwrites instead (test :490-507)
    b = dsa.dynamicsparse(fill_mode=False, binding=oracle)
    b.set_batch(row, col, val)
    last = {}
    for r, c, x in zip(row, col, val):
        last[(r, c)] = x
    keys = list(last)[:3000]
    assert np.array_equal(b.get_batch([k[0] for k in keys], [k[1] for k in keys]), np.array([last[k] for k in keys]))


def test_negative_and_huge_keys(dsa, oracle):
    v = dsa.dynamicsparsevec([-5, 10 ** 15, 3, -(10 ** 12)], [1.0, 2.0, 3.0, 4.0], binding=oracle)
    assert [k for k, _ in v] == [-(10 ** 12), -5, 3, 10 ** 15]
    v[-7] = 9.0
    assert v[-7] == 9.0 and v[10 ** 15] == 2.0 and len(v) == 10 ** 15
    a = dsa.dynamicsparse([1, 2, 3], [5, -2, 10 ** 13], [1.0, 2.0, 3.0], binding=oracle)
    a[7, -9] = 4.0                                       # new column in front of everything (sparsematrix.jl:251)
    assert a[2, -2] == 2.0 and a[7, -9] == 4.0 and a[3, 10 ** 13] == 3.0
    assert a.col_view(-9) == [(7, 4.0)] and a.row_view(3) == [(10 ** 13, 3.0)]


def test_row_and_column_slices(dsa, oracle):
    """test/functional/sparsematrix.jl:212-245 (A.5): m[2, :] / m[:, 2] as dynamic sparse vectors."""
    J = [1, 1, 1, 2, 2, 2, 3, 3, 3]
    I = [1, 2, 3, 2, 6, 7, 1, 6, 8]
    V = [2, 3, 4, 2, 4, 5, 3, 5, 7]
    a = dsa.dynamicsparse(I, J, V, binding=oracle)
    a[1, 1] = 4; a[1, 2] = 3; a[3, 1] = 0; a[4, 2] = 1
    row = a.row_slice(2)
    assert row.nnz() == 2 and [row[j] for j in (1, 2, 3)] == [3.0, 2.0, 0.0]
    col = a.col_slice(2)
    assert col.nnz() == 5 and [col[i] for i in range(1, 9)] == [3.0, 2.0, 0.0, 1.0, 0.0, 4.0, 5.0, 0.0]
    assert a.col_slice(99).nnz() == 0 and a.row_slice(5).nnz() == 0


def _vec_math_case(seed, n=25, keyspace=100):
    """the shape of addition() / subtraction() of test/functional/math.jl:53-94: 25 random keys in 1:100 (duplicates combined by +),
    integer values 1:10 — drawn from our splitmix64"""
    from util import splitmix_array
    k = 1 + (splitmix_array(seed, n) % np.uint64(keyspace)).astype(np.int64)
    v = (1 + splitmix_array(seed + 1000, n) % np.uint64(10)).astype(np.float64)
    return k, v


def _dense(k, v, n):
    d = np.zeros(n + 1)
    np.add.at(d, np.asarray(k, dtype=np.int64), v)
    return d


@pytest.mark.parametrize("seed", range(5))
def test_vector_addition_subtraction_negation(dsa, oracle, seed):
    """test/functional/math.jl:53-94: dyn_vec1 + dyn_vec2 == sparsevec sum, dyn_vec1 - dyn_vec2, -dyn_vec (value comparison)."""
    k1, v1 = _vec_math_case(10 + seed)
    k2, v2 = _vec_math_case(20 + seed)
    a = dsa.dynamicsparsevec(k1, v1, n=100, binding=oracle)
    b = dsa.dynamicsparsevec(k2, v2, n=100, binding=oracle)
    d1, d2 = _dense(k1, v1, 100), _dense(k2, v2, 100)
    for (ks, vs), want in (((a + b), d1 + d2), ((a - b), d1 - d2), ((-a), -d1), ((a - a), d1 * 0)):
        assert np.all(np.diff(ks) > 0)
        assert np.array_equal(_dense(ks, vs, 100), want)
    ks, vs = a - a                                    # entries stored on both sides cancel: dropped
    assert len(ks) == 0


def test_vector_equality_and_filter(dsa, oracle):
    """v1 == v2 compares length, stored count and the stored tuples in order — not the slot layout (src/vector.jl:85-87,
    src/pma.jl:236-266; test/functional/sparsevector.jl:64-77); filter builds a new vector (src/pma.jl:224-234, sparsevector.jl:81-86)."""
    k = np.arange(1, 401, dtype=np.int64) * 3
    v = np.arange(1, 401, dtype=np.float64)
    a = dsa.dynamicsparsevec(k, v, binding=oracle)
    b = dsa.dynamicsparsevec(k[:100], v[:100], binding=oracle)
    assert not (a == b)
    b.set_batch(k[100:], v[100:])                     # same content, other history -> other layout
    assert not np.array_equal(a.export_layout()[2], b.export_layout()[2])
    assert a == b and a == a
    b[7] = 1.0
    assert not (a == b)
    b[7] = 0.0
    assert a == b
    b[2000] = 1.0
    b[2000] = 0.0                                     # length(b) grew: no longer equal until shrink_size!
    assert not (a == b)
    b.shrink_size()
    assert a == b
    b[3] = float("nan")
    c = dsa.dynamicsparsevec(*b.nonzeros(), binding=oracle)
    assert b == b and not (b == c)                    # === short cut; NaN != NaN element-wise
    f = a.filter(lambda e: e[0] % 2 == 0 and e[1] > 10)
    fk, fv = f.nonzeros()
    sel = (k % 2 == 0) & (v > 10)
    assert np.array_equal(fk, k[sel]) and np.array_equal(fv, v[sel]) and len(f) == int(k[sel].max())


def test_oracle_layout_import_export_round_trip(dsa, oracle):
    """ora_vec_import_layout / ora_pcsc_import_layout (the checker's side of dsa_*_import_layout): a structure restored from an
    exported layout has the same scalars and slots, and goes on behaving like the original."""
    import numpy as np
    from util import layouts_equal, check_semaphores
    rng = np.random.default_rng(3)
    keys = np.sort(rng.choice(50000, size=3000, replace=False)) + 1
    a = dsa.dynamicsparsevec(keys, rng.random(3000) + 1.0, binding=oracle)
    a.set_batch(rng.choice(50000, size=2000) + 1, np.where(rng.random(2000) < 0.3, 0.0, 2.0))
    k, v, o = a.export_layout()
    b = dsa.import_vector_layout(k, v, o, a.info()["segment_capacity"], n=len(a), binding=oracle)
    for f in ("capacity", "segment_capacity", "nb_segments", "nb_elements", "height"):
        assert a.info()[f] == b.info()[f], f
    assert layouts_equal(a.export_layout(), b.export_layout()) and a == b
    ops_k, ops_v = rng.choice(50000, size=4000) + 1, np.where(rng.random(4000) < 0.4, 0.0, 3.0)
    a.set_batch(ops_k, ops_v)
    b.set_batch(ops_k, ops_v)
    assert layouts_equal(a.export_layout(), b.export_layout()) and a.info()["capacity"] == b.info()["capacity"]
    rows = [np.sort(rng.choice(900, size=int(c), replace=False)) + 1 for c in rng.integers(0, 30, 40)]
    p = dsa.packedcsc(rows, [rng.random(len(r)) + 1.0 for r in rows], binding=oracle)
    p.deletepartition(7)
    pk, pv, po, ps = p.export_layout()
    q = dsa.import_packedcsc_layout(pk, pv, po, p.info()["segment_capacity"], ps, binding=oracle)
    assert q.nbpartitions() == p.nbpartitions() and q.nnz() == p.nnz()
    for t in (p, q):
        t[5, 12] = 4.0
        t[901, 3] = 1.5
    lp, lq = p.export_layout(), q.export_layout()
    assert layouts_equal(lp[:3], lq[:3]) and np.array_equal(lp[3], lq[3])
    check_semaphores(lq[0], lq[1], lq[2], lq[3])


def test_raw_find_agrees_with_a_linear_scan_when_the_range_is_key_partitioned(oracle):
    """find (src/finds.jl:29-57) on raw arrays against a brute-force statement of its contract: the cell holding the key if it is
    stored in [from, to]; else the last cell of the range with a smaller key; else the nearest cell left of `from`; else (0, nothing)."""
    import numpy as np
    from rawhooks import Raw, key_partitioned, random_partitioned_array
    rng = np.random.default_rng(12)
    ref = Raw(oracle)
    for _ in range(60):
        n = int(rng.choice([5, 17, 64, 200]))
        k, v, o, _s = random_partitioned_array(rng, n, float(rng.choice([0.2, 0.6, 0.95])), 0, key_hi=3 * n)
        for _q in range(20):
            frm = int(rng.integers(1, n + 1)); to = int(rng.integers(frm - 1, n + 1)); key = int(rng.integers(0, 3 * n + 2))
            assert key_partitioned(k, o, key, frm, to)
            pos, elem = ref.find(k, v, o, key, frm, to)
            occ = [i + 1 for i in range(n) if o[i]]
            inr = [p for p in occ if frm <= p <= to]
            hit = [p for p in inr if k[p - 1] == key]
            if hit:
                exp = hit[0]
            else:
                smaller = [p for p in inr if k[p - 1] < key]
                left = [p for p in occ if p < frm]
                exp = smaller[-1] if smaller else (left[-1] if left else 0)
            assert pos == exp, (key, frm, to, pos, exp)
            assert (elem is None) == (exp == 0)
