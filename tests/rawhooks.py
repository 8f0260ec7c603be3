"""Uniform access to the raw slot-array primitives of the two libraries: the CPU oracle's `ora_raw_*` (oracle/oracle_capi.cpp)
and the HIP library's parity hooks `dsa_dbg_raw_*` (include/dsa.h), which run the DEVICE code on a caller-supplied slot array.
A raw array is a list of `None | [key, value]` (the shape of the reference's unit tests) or a (keys, vals, occ) triple."""
import ctypes as C

import numpy as np

P_I64 = C.POINTER(C.c_int64)
P_F64 = C.POINTER(C.c_double)
P_U8 = C.POINTER(C.c_uint8)

BLOCK, WAVE, GRID = 0, 1, 2
ENGINE_NAMES = {BLOCK: "block", WAVE: "wave", GRID: "grid"}


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


def _p(a, t):
    return a.ctypes.data_as(t)


class Raw:
    """engine = None: the oracle; otherwise the HIP parity hooks on that engine (fast = wave-parallel K-find)."""

    def __init__(self, lib, engine=None, fast=0):
        self.b = lib
        self.engine = engine
        self.fast = fast

    @property
    def name(self):
        return "oracle" if self.engine is None else f"hip-{ENGINE_NAMES[self.engine]}{'-fast' if self.fast else ''}"

    def find(self, k, v, o, key, frm, to):
        pos, has, fk, fv = C.c_int64(), C.c_int32(), C.c_int64(), C.c_double()
        if self.engine is None:
            rc = self.b.lib.ora_raw_find(_p(k, P_I64), _p(v, P_F64), _p(o, P_U8), C.c_int64(len(o)), C.c_int64(key), C.c_int64(frm),
                                         C.c_int64(to), C.byref(pos), C.byref(has), C.byref(fk), C.byref(fv))
            assert rc == 0
        else:
            self.b.call("dbg_raw_find", _p(k, P_I64), _p(v, P_F64), _p(o, P_U8), len(o), key, frm, to, self.engine, self.fast,
                        C.byref(pos), C.byref(has), C.byref(fk), C.byref(fv))
        return pos.value, ([fk.value, fv.value] if has.value else None)

    def insert(self, k, v, o, key, val, frm, to, sems=None):
        """in place; returns (status, pos, is_new)"""
        pos, isnew = C.c_int64(), C.c_int32()
        sp = _p(sems, P_I64) if sems is not None else None
        ns = len(sems) if sems is not None else 0
        if self.engine is None:
            rc = self.b.lib.ora_raw_insert(_p(k, P_I64), _p(v, P_F64), _p(o, P_U8), C.c_int64(len(o)), C.c_int64(key), C.c_double(val),
                                           C.c_int64(frm), C.c_int64(to), sp, C.c_int64(ns), C.byref(pos), C.byref(isnew))
        else:
            rc = self.b._dbg_raw_insert(_p(k, P_I64), _p(v, P_F64), _p(o, P_U8), len(o), key, float(val), frm, to, sp, ns, self.engine,
                                        self.fast, C.byref(pos), C.byref(isnew))
        return rc, pos.value, bool(isnew.value)

    def delete(self, k, v, o, key, frm, to):
        pos, deleted = C.c_int64(), C.c_int32()
        if self.engine is None:
            rc = self.b.lib.ora_raw_delete(_p(k, P_I64), _p(v, P_F64), _p(o, P_U8), C.c_int64(len(o)), C.c_int64(key), C.c_int64(frm),
                                           C.c_int64(to), C.byref(pos), C.byref(deleted))
            assert rc == 0
        else:
            self.b.call("dbg_raw_delete", _p(k, P_I64), _p(v, P_F64), _p(o, P_U8), len(o), key, frm, to, self.engine, self.fast,
                        C.byref(pos), C.byref(deleted))
        return pos.value, bool(deleted.value)

    def purge(self, k, v, o, frm, to):
        mid, nb = C.c_int64(), C.c_int64()
        if self.engine is None:
            rc = self.b.lib.ora_raw_purge(_p(k, P_I64), _p(v, P_F64), _p(o, P_U8), C.c_int64(len(o)), C.c_int64(frm), C.c_int64(to),
                                          C.byref(mid), C.byref(nb))
            assert rc == 0
        else:
            self.b.call("dbg_raw_purge", _p(k, P_I64), _p(v, P_F64), _p(o, P_U8), len(o), frm, to, C.byref(mid), C.byref(nb))
        return mid.value, nb.value

    def rebalance(self, k, v, o, ws, we, sems=None):
        """pack! + spread! of [ws, we] in place (the cell count is taken from occ)"""
        sp = _p(sems, P_I64) if sems is not None else None
        ns = len(sems) if sems is not None else 0
        if self.engine is None:
            m = int(o[ws - 1:we].sum())
            rc = self.b.lib.ora_raw_pack_spread(_p(k, P_I64), _p(v, P_F64), _p(o, P_U8), C.c_int64(len(o)), C.c_int64(ws), C.c_int64(we),
                                                C.c_int64(m), sp, C.c_int64(ns), C.c_int32(1), C.c_int32(1))
            assert rc == 0
        else:
            self.b.call("dbg_raw_rebalance", _p(k, P_I64), _p(v, P_F64), _p(o, P_U8), len(o), ws, we, sp, ns, self.engine)


def key_partitioned(k, o, key, frm, to):
    """the precondition under which the wave-parallel K-find must agree with the bisection (csrc/find_dev.h): inside [frm, to]
    every stored key < `key` precedes every stored key > `key`, and `key` itself is stored at most once, between them"""
    s = [int(k[i]) for i in range(max(frm, 1) - 1, min(to, len(o))) if o[i]]
    if s.count(key) > 1:
        return False
    p = next((i for i, a in enumerate(s) if a >= key), len(s))
    if not all(a < key for a in s[:p]):
        return False
    rest = s[p:]
    if rest and rest[0] == key:
        rest = rest[1:]
    return all(a > key for a in rest)


def random_partitioned_array(rng, len_, density, nparts, key_hi=10 ** 6, wide=False):
    """a raw array in PackedCSC shape: `nparts` partitions, each a semaphore cell (0, id) followed by ascending keys, spread at
    random over `len_` slots (the shape of the reference's partitioned_array_factory, test/utils.jl:41-66); returns k, v, o, sems"""
    n = max(nparts, int(len_ * density))
    n = min(n, len_)
    pos = np.sort(rng.choice(len_, size=n, replace=False))
    starts = np.sort(rng.choice(n, size=nparts, replace=False)) if nparts > 0 else np.array([], dtype=np.int64)
    if nparts > 0:
        starts[0] = 0
        starts = np.unique(starts)
    k = np.zeros(len_, dtype=np.int64)
    v = np.zeros(len_, dtype=np.float64)
    o = np.zeros(len_, dtype=np.uint8)
    sems = np.zeros(len(starts), dtype=np.int64)
    bounds = list(starts) + [n]
    off = (1 << 40) if wide else 0
    for pid in range(len(starts)):
        a, b = bounds[pid], bounds[pid + 1]
        cnt = b - a - 1
        keys = np.sort(rng.choice(key_hi, size=cnt, replace=False)) + 1 + off if cnt > 0 else np.array([], dtype=np.int64)
        sems[pid] = pos[a] + 1
        k[pos[a]], v[pos[a]], o[pos[a]] = 0, float(pid + 1), 1
        for j in range(cnt):
            s = pos[a + 1 + j]
            k[s], v[s], o[s] = keys[j], float(rng.integers(1, 1000)), 1
    if nparts == 0:
        keys = np.sort(rng.choice(key_hi, size=n, replace=False)) + 1 + off
        for j in range(n):
            k[pos[j]], v[pos[j]], o[pos[j]] = keys[j], float(rng.integers(1, 1000)), 1
    return k, v, o, (sems if nparts > 0 else None)
