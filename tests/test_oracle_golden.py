"""CPU tests: the oracle against the reference's own known-answer vectors
(tests/golden/reference_cases.json, transcribed from the reference's test suite)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from scenario import CODES, run_scenario

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_cases.json")) as f:
    CASES = json.load(f)

P_I64 = C.POINTER(C.c_int64)
P_F64 = C.POINTER(C.c_double)
P_U8 = C.POINTER(C.c_uint8)


def to_arrays(slots):
    n = len(slots)
    k = np.zeros(n, dtype=np.int64)
    v = np.zeros(n, dtype=np.float64)
    o = np.zeros(n, dtype=np.uint8)
    for i, s in enumerate(slots):
        if s is not None:
            k[i], v[i], o[i] = s[0], s[1], 1
    return k, v, o


def from_arrays(k, v, o):
    return [[int(k[i]), float(v[i])] if o[i] else None for i in range(len(o))]


def norm(slots):
    return [None if s is None else [int(s[0]), float(s[1])] for s in slots]


@pytest.mark.parametrize("case", CASES["find"], ids=lambda c: c["ref"])
def test_find_vectors(oracle, case):
    k, v, o = to_arrays(case["array"])
    lib = oracle.lib
    for key, exp_pos, exp_elem in case["queries"]:
        pos, has, fk, fv = C.c_int64(), C.c_int32(), C.c_int64(), C.c_double()
        rc = lib.ora_raw_find(k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64), o.ctypes.data_as(P_U8), C.c_int64(len(o)),
                              C.c_int64(key), C.c_int64(case["frm"]), C.c_int64(case["to"]), C.byref(pos), C.byref(has),
                              C.byref(fk), C.byref(fv))
        assert rc == 0
        assert pos.value == exp_pos, (key, pos.value, exp_pos)
        if exp_elem is None:
            assert has.value == 0
        else:
            assert has.value == 1 and [fk.value, fv.value] == [exp_elem[0], float(exp_elem[1])]


@pytest.mark.parametrize("case", CASES["arrays_equal"], ids=lambda c: c["ref"])
def test_arrays_equal_vectors(oracle, case):
    (k1, v1, o1), (k2, v2, o2) = to_arrays(case["a"]), to_arrays(case["b"])
    out = C.c_int32(-1)
    rc = oracle.lib.ora_raw_arrays_equal(k1.ctypes.data_as(P_I64), v1.ctypes.data_as(P_F64), o1.ctypes.data_as(P_U8), C.c_int64(len(o1)),
                                         k2.ctypes.data_as(P_I64), v2.ctypes.data_as(P_F64), o2.ctypes.data_as(P_U8), C.c_int64(len(o2)),
                                         C.byref(out))
    assert rc == 0 and bool(out.value) is case["expect"]


@pytest.mark.parametrize("case", CASES["insert"], ids=lambda c: c["ref"])
def test_insert_vectors(oracle, case):
    k, v, o = to_arrays(case["array"])
    lib = oracle.lib
    for st in case["steps"]:
        pos, isnew = C.c_int64(), C.c_int32()
        rc = lib.ora_raw_insert(k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64), o.ctypes.data_as(P_U8), C.c_int64(len(o)),
                                C.c_int64(st["key"]), C.c_double(st["val"]), C.c_int64(st["frm"]), C.c_int64(st["to"]),
                                None, C.c_int64(0), C.byref(pos), C.byref(isnew))
        if "error" in st:
            assert rc == CODES[st["error"]]
        else:
            assert rc == 0
            assert from_arrays(k, v, o) == norm(st["expect"])


def test_delete_purge_vectors(oracle):
    case = CASES["delete"]
    k, v, o = to_arrays(case["array"])
    lib = oracle.lib
    for st in case["steps"]:
        if st["op"] == "delete":
            pos, deleted = C.c_int64(), C.c_int32()
            rc = lib.ora_raw_delete(k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64), o.ctypes.data_as(P_U8), C.c_int64(len(o)),
                                    C.c_int64(st["key"]), C.c_int64(1), C.c_int64(len(o)), C.byref(pos), C.byref(deleted))
            assert rc == 0 and [pos.value, bool(deleted.value)] == st["out"]
        else:
            mid, nb = C.c_int64(), C.c_int64()
            rc = lib.ora_raw_purge(k.ctypes.data_as(P_I64), v.ctypes.data_as(P_F64), o.ctypes.data_as(P_U8), C.c_int64(len(o)),
                                   C.c_int64(st["frm"]), C.c_int64(st["to"]), C.byref(mid), C.byref(nb))
            assert rc == 0 and [mid.value, nb.value] == st["out"]
        assert from_arrays(k, v, o) == norm(st["expect"])


@pytest.mark.parametrize("sc", CASES["scenarios"], ids=lambda s: s["name"])
def test_reference_scenarios(dsa, oracle, sc):
    run_scenario(dsa, oracle, sc)
