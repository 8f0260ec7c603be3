"""Interpreter for the scenario scripts of tests/golden/reference_cases.json.
Runs a scenario against any Binding (CPU oracle or the HIP library)."""
import numpy as np
import pytest

from util import check_key_order, check_semaphores

CODES = dict(EARG=1, EBOUNDS=2, EDELETED=3, EFULL=4, EMODE=5, EASSERT=6, EHIP=7, ECAP=8, EKEY=9)


def _check_pcsc_layout(keys, vals, occ, sems, expect_live=None):
    live = check_semaphores(keys, vals, occ, sems)
    check_key_order(keys, occ)
    if expect_live is not None:
        assert live == expect_live


def run_vector(dsa, b, sc):
    c = sc["create"]
    v = dsa.dynamicsparsevec(c["I"], c["V"], combine=c.get("combine", "+"), n=c.get("n"), binding=b)
    for st in sc["steps"]:
        op = st[0]
        if op == "get":
            assert v[st[1]] == st[2], st
        elif op == "set":
            v[st[1]] = st[2]
        elif op == "len":
            assert len(v) == st[1], st
        elif op == "nnz":
            assert v.nnz() == st[1], st
        elif op == "capacity":
            assert v.info()["capacity"] == st[1], st
        elif op == "iter":
            assert [list(x) for x in v] == st[1], st
        elif op == "filter_even_keys_equals":      # filter(e -> e[1] % 2 == 0, v) == dynamicsparsevec(I, V)
            f = v.filter(lambda e: e[0] % 2 == 0)
            assert f == dsa.dynamicsparsevec(st[1], st[2], binding=b), st
            assert [list(x) for x in f] == [[k, float(x)] for k, x in zip(st[1], st[2])], st
        else:
            raise AssertionError(op)
    return v


def run_vector_pair(dsa, b, sc):
    va = dsa.dynamicsparsevec(sc["a"]["I"], sc["a"]["V"], binding=b)
    vb = dsa.dynamicsparsevec(sc["b"]["I"], sc["b"]["V"], binding=b)
    for st in sc["steps"]:
        if st[0] == "expect_equal":
            assert (va == vb) is st[1]
        elif st[0] == "b_set":
            vb[st[1]] = st[2]
        elif st[0] == "shrink_both":
            va.shrink_size()
            vb.shrink_size()


def _pcsc_step(dsa, p, st):
    op = st[0]
    if op == "set":
        p[st[1], st[2]] = st[3]
    elif op == "add":
        p[st[1], st[2]] = p[st[1], st[2]] + st[3]
    elif op == "get":
        assert p[st[1], st[2]] == st[3], st
    elif op == "nnz":
        assert p.nnz() == st[1], st
    elif op == "nbpartitions":
        assert p.nbpartitions() == st[1], st
    elif op == "deletepartition":
        p.deletepartition(st[1])
    elif op == "dense":
        for i, row in enumerate(st[1], start=1):
            for j, x in enumerate(row, start=1):
                assert p[i, j] == x, (i, j, x)
    elif op == "check_invariants":
        k, v, o, s = p.export_layout()
        _check_pcsc_layout(k, v, o, s, st[1] if len(st) > 1 else None)
    elif op == "expect_error":
        with pytest.raises(dsa.DsaError) as ei:
            _pcsc_step(dsa, p, st[2])
        assert ei.value.code == CODES[st[1]], (ei.value.code, st)
    else:
        raise AssertionError(op)


def run_pcsc(dsa, b, sc):
    c = sc["create"]
    p = dsa.packedcsc(c["row_keys"], c["values"], binding=b)
    for st in sc["steps"]:
        _pcsc_step(dsa, p, st)
    return p


def _mat_step(dsa, a, st):
    op = st[0]
    if op == "set":
        a[st[1], st[2]] = st[3]
    elif op == "add":
        a[st[1], st[2]] = a[st[1], st[2]] + st[3]
    elif op == "get":
        assert a[st[1], st[2]] == st[3], st
    elif op == "nnz":
        assert a.nnz() == st[1], st
    elif op == "size":
        assert a.size() == (st[1], st[2]), (a.size(), st)
    elif op == "nbpartitions":
        assert a.nbpartitions(st[1]) == st[2], (a.nbpartitions(st[1]), st)
    elif op == "deletecolumn":
        a.deletecolumn(st[1])
    elif op == "deleterow":
        a.deleterow(st[1])
    elif op == "addrow":
        a.addrow(st[1], st[2], st[3])
    elif op == "closefillmode":
        a.closefillmode()
    elif op == "col_view":
        assert [list(x) for x in a.col_view(st[1])] == st[2], st
    elif op == "row_view":
        assert [list(x) for x in a.row_view(st[1])] == st[2], st
    elif op == "dense":
        for i, row in enumerate(st[1], start=1):
            for j, x in enumerate(row, start=1):
                assert a[i, j] == x, (i, j, x)
    elif op == "dense_rowmajor":   # matrix.rowmajor[j, i]  (test/functional/sparsematrix.jl:449)
        for i, row in enumerate(st[1], start=1):
            got = dict(a.row_view(i))
            for j, x in enumerate(row, start=1):
                assert got.get(j, 0.0) == x, (i, j, x)
    elif op == "mul":
        _, transpose, xi, xv, expect, zeros = st
        yi, yv = a.mul((xi, xv), transpose=bool(transpose))
        got = dict(zip(yi.tolist(), yv.tolist()))
        assert list(yi) == sorted(yi.tolist())
        for k, val in expect.items():
            assert got.get(int(k), 0.0) == val, (k, got)
        for k in zeros:
            assert got.get(k, 0.0) == 0.0, (k, got)
        # dense front end gives the same numbers
        m, n = a.size()
        nx = max(max(xi), m, n)
        x = np.zeros(nx)
        x[np.array(xi) - 1] = xv
        y = a.mul(x, transpose=bool(transpose))
        for k, val in got.items():
            if 1 <= k <= len(y):
                assert y[k - 1] == val
    elif op == "check_invariants":
        for o in (0, 1):
            L = a.export_layout(o)
            exp = st[1 + o] if len(st) > 1 + o else None
            _check_pcsc_layout(L["keys"], L["vals"], L["occ"], L["semaphores"], exp)
            assert np.array_equal(L["col_live"] != 0, L["semaphores"] != 0)
            ck = L["col_keys"][L["col_live"] != 0]
            assert np.all(np.diff(ck) > 0)
    elif op == "expect_error":
        with pytest.raises(dsa.DsaError) as ei:
            _mat_step(dsa, a, st[2])
        assert ei.value.code == CODES[st[1]], (ei.value.code, st)
    else:
        raise AssertionError(op)


def run_matrix(dsa, b, sc):
    c = sc["create"]
    if "I" in c:
        a = dsa.dynamicsparse(c["I"], c["J"], c["V"], binding=b)
    else:
        a = dsa.dynamicsparse(fill_mode=c["fill_mode"], binding=b)
    for st in sc["steps"]:
        _mat_step(dsa, a, st)
    return a


def run_scenario(dsa, b, sc):
    return dict(vector=run_vector, vector_pair=run_vector_pair, pcsc=run_pcsc, matrix=run_matrix)[sc["kind"]](dsa, b, sc)
