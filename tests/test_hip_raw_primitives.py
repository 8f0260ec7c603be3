"""GPU parity tests of the slot-array primitives THEMSELVES (run with `-m gpu`): the device code of find / insert! / delete! /
purge! / pack! + spread! is driven through the parity hooks of the C ABI (`dsa_dbg_raw_*`, include/dsa.h) on raw slot arrays.

1. Every slot-level golden vector the reference's own unit tests hold — test/unit/finds.jl:4-107, test/unit/writes.jl:5-70,
   test/unit/comparison.jl:2-10, transcribed as data in tests/golden/reference_cases.json — runs on the DEVICE implementations
   (both engines: the sequencer's workgroup primitives and the wave-level primitives of the batch-parallel rounds; K-find in its
   literal bisection form and, where the searched range is key-partitioned, in its wave-parallel 64-ary form).
2. The same primitives against the CPU oracle on seeded random arrays: long shifts in both directions with semaphore fix-up
   (the shape of test/unit/moves.jl:45-117), purges, windows of every size on the three rebalance engines incl. the fp64
   gap-placement edge cases of SURVEY App. A.3.
3. Layouts restored with dsa_*_import_layout behave like the structures they were exported from."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from rawhooks import (BLOCK, GRID, WAVE, Raw, from_arrays, key_partitioned, norm, random_partitioned_array, to_arrays)
from scenario import CODES
from util import check_key_order, check_semaphores, layouts_equal

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_cases.json")) as f:
    CASES = json.load(f)

ENGINES = [BLOCK, WAVE]


# ------------------------------------------------------------------ 1. the reference's golden vectors on the device
@pytest.mark.parametrize("engine", ENGINES, ids=["block", "wave"])
@pytest.mark.parametrize("case", CASES["find"], ids=lambda c: c["ref"])
def test_reference_find_vectors_on_device(hip, case, engine):
    k, v, o = to_arrays(case["array"])
    n_fast = 0
    for key, exp_pos, exp_elem in case["queries"]:
        exp = (exp_pos, None if exp_elem is None else [exp_elem[0], float(exp_elem[1])])
        assert Raw(hip, engine, 0).find(k, v, o, key, case["frm"], case["to"]) == exp, (key, "bisection")
        if key_partitioned(k, o, key, case["frm"], case["to"]):
            assert Raw(hip, engine, 1).find(k, v, o, key, case["frm"], case["to"]) == exp, (key, "64-ary")
            n_fast += 1
    # the narrative cases of finds.jl:62-107 search ranges that are NOT key-partitioned: the bisection alone is the contract there
    if case["ref"] in ("test/unit/finds.jl:4-23", "test/unit/finds.jl:26-48"):
        assert n_fast == len(case["queries"])


@pytest.mark.parametrize("engine", ENGINES, ids=["block", "wave"])
@pytest.mark.parametrize("fast", [0, 1], ids=["bisection", "64ary"])
@pytest.mark.parametrize("case", CASES["insert"], ids=lambda c: c["ref"])
def test_reference_insert_vectors_on_device(hip, case, engine, fast):
    k, v, o = to_arrays(case["array"])
    for st in case["steps"]:
        use_fast = fast and key_partitioned(k, o, st["key"], st["frm"], st["to"])
        rc, pos, is_new = Raw(hip, engine, int(use_fast)).insert(k, v, o, st["key"], st["val"], st["frm"], st["to"])
        if "error" in st:
            assert rc == CODES[st["error"]]            # "No empty cell to insert a new element."  src/writes.jl:39
        else:
            assert rc == 0
            assert from_arrays(k, v, o) == norm(st["expect"]), st


@pytest.mark.parametrize("engine", ENGINES, ids=["block", "wave"])
@pytest.mark.parametrize("fast", [0, 1], ids=["bisection", "64ary"])
def test_reference_delete_purge_vectors_on_device(hip, engine, fast):
    case = CASES["delete"]
    k, v, o = to_arrays(case["array"])
    for st in case["steps"]:
        if st["op"] == "delete":
            use_fast = fast and key_partitioned(k, o, st["key"], 1, len(o))
            assert list(Raw(hip, engine, int(use_fast)).delete(k, v, o, st["key"], 1, len(o))) == st["out"]
        else:
            assert list(Raw(hip, BLOCK).purge(k, v, o, st["frm"], st["to"])) == st["out"]
        assert from_arrays(k, v, o) == norm(st["expect"])


@pytest.mark.parametrize("case", CASES["arrays_equal"], ids=lambda c: c["ref"])
def test_reference_arrays_equal_vectors_on_device(dsa, hip, case):
    """_arrays_equal (src/pma.jl:236-260) compares stored tuples, not slots: the reference's two raw arrays, padded with empty slots
    to a power-of-two capacity, are restored as vectors and compared by the device kernel behind `==`."""
    def vec(slots):
        cap = 4
        while cap < len(slots):
            cap *= 2
        k, v, o = to_arrays(list(slots) + [None] * (cap - len(slots)))
        return dsa.import_vector_layout(k, v, o, 2, n=100, binding=hip)
    a, b = vec(case["a"]), vec(case["b"])
    assert (a == b) is case["expect"]
    assert (b == a) is case["expect"]


# ------------------------------------------------------------------ 2. the primitives against the oracle on random arrays
def _same(x, y):
    return layouts_equal(x, y)


@pytest.mark.parametrize("engine", ENGINES, ids=["block", "wave"])
@pytest.mark.parametrize("wide", [False, True], ids=["keys32", "keys64"])
def test_insert_shifts_with_semaphores_match_oracle(hip, oracle, engine, wide):
    """insert! into dense regions of a partitioned array: the shift to the nearest gap on the right, else on the left
    (src/writes.jl:26-43), runs of up to ~700 cells — several chunks of the 256-thread / 64-lane shift loops — carrying
    semaphore cells whose table entries must follow (src/moves.jl:26-42,69-85)."""
    rng = np.random.default_rng(20 + engine + 2 * wide)
    ref = Raw(oracle)
    dev = Raw(hip, engine, 1)
    n_left = n_right = n_over = n_sem_moved = 0
    for trial in range(40):
        len_ = int(rng.choice([64, 200, 777, 1500, 4096]))
        density = float(rng.choice([0.5, 0.8, 0.97, 0.995]))
        nparts = int(rng.integers(1, max(2, len_ // 24)))
        k, v, o, sems = random_partitioned_array(rng, len_, density, nparts, wide=wide)
        off = (1 << 40) if wide else 0
        # in every third trial the slots behind the last semaphore are all occupied: an insert there finds no gap on its right
        # and takes the left branch of _insert! (src/writes.jl:33-37), shifting across the semaphores in front of it
        if trial % 3 == 0:
            last = int(sems[-1])
            idx = np.arange(last, len_)
            o[idx] = 1
            k[idx] = np.arange(1, len(idx) + 1) * 3 + off
            v[idx] = 1.0
        for step in range(12):
            if o.all():
                break
            pid = int(rng.integers(0, len(sems)))
            frm = int(sems[pid]) + 1
            to = int(sems[pid + 1]) - 1 if pid + 1 < len(sems) else len_
            stored = [int(k[i]) for i in range(frm - 1, to) if o[i]]
            if stored and rng.random() < 0.2:
                key = int(rng.choice(stored))                      # overwrite
            else:
                key = int(rng.integers(1, 3 * 10 ** 6)) * 3 + 1 + ((1 << 40) if wide else 0)
            val = float(rng.integers(1, 1000))
            assert key_partitioned(k, o, key, frm, to)
            a = (k.copy(), v.copy(), o.copy(), sems.copy())
            b = (k.copy(), v.copy(), o.copy(), sems.copy())
            ra = ref.insert(a[0], a[1], a[2], key, val, frm, to, a[3])
            rb = dev.insert(b[0], b[1], b[2], key, val, frm, to, b[3])
            assert ra == rb, (trial, step, ra, rb)
            assert ra[0] == 0
            assert _same(a[:3], b[:3]), (trial, step)
            assert np.array_equal(a[3], b[3]), (trial, step)
            if not ra[2]:
                n_over += 1
            elif not np.array_equal(o[:ra[1] - 1], a[2][:ra[1] - 1]):
                n_left += 1                                        # a bit LEFT of the insertion point changed: the left branch
            else:
                n_right += 1
            n_sem_moved += int((a[3] != sems).sum())
            k, v, o, sems = a
            check_semaphores(k, v, o, sems)
    assert n_left > 5 and n_right > 50 and n_over > 5 and n_sem_moved > 20, (n_left, n_right, n_over, n_sem_moved)


@pytest.mark.parametrize("engine", ENGINES, ids=["block", "wave"])
def test_find_delete_purge_match_oracle_on_random_arrays(hip, oracle, engine):
    rng = np.random.default_rng(7 + engine)
    ref = Raw(oracle)
    for trial in range(25):
        len_ = int(rng.choice([7, 64, 65, 300, 1000, 5000]))
        k, v, o, _ = random_partitioned_array(rng, len_, float(rng.choice([0.1, 0.5, 0.9])), 0, key_hi=4 * len_)
        for q in range(12):
            frm = int(rng.integers(1, len_ + 1))
            to = int(rng.integers(frm - 1, len_ + 1))
            key = int(rng.integers(0, 4 * len_ + 2))
            exp = ref.find(k, v, o, key, frm, to)
            assert Raw(hip, engine, 0).find(k, v, o, key, frm, to) == exp, (trial, q, "bisection")
            assert key_partitioned(k, o, key, frm, to)
            assert Raw(hip, engine, 1).find(k, v, o, key, frm, to) == exp, (trial, q, "64-ary")
        for q in range(4):
            stored = k[o.astype(bool)]
            key = int(rng.choice(stored)) if len(stored) and rng.random() < 0.7 else int(rng.integers(0, 4 * len_ + 2))
            a, b = (k.copy(), v.copy(), o.copy()), (k.copy(), v.copy(), o.copy())
            assert ref.delete(*a, key, 1, len_) == Raw(hip, engine, q & 1).delete(*b, key, 1, len_)
            assert _same(a, b)
            k, v, o = a
        frm = int(rng.integers(1, len_ + 1))
        to = int(rng.integers(frm - 1, len_ + 1))
        a, b = (k.copy(), v.copy(), o.copy()), (k.copy(), v.copy(), o.copy())
        assert ref.purge(*a, frm, to) == Raw(hip, BLOCK).purge(*b, frm, to)
        assert _same(a, b)


# (W, m) with floor(fl(k * fl(W / E))) != floor(k W / E) for some k, empty / full / single-cell windows
FP_EDGE = [(64, 15), (128, 25), (128, 21), (64, 1), (64, 63), (64, 64), (64, 0), (128, 127), (256, 1)]


@pytest.mark.parametrize("engine", [BLOCK, WAVE, GRID], ids=["block", "wave", "grid"])
def test_window_rebalance_matches_oracle(hip, oracle, engine):
    """pack! + spread! (src/moves.jl:94-171) of interior windows of every size each engine serves, cells anywhere in the window,
    semaphore cells among them; slots outside the window, and semaphores outside it, must not change."""
    rng = np.random.default_rng(100 + engine)
    ref, dev = Raw(oracle), Raw(hip, engine)
    wmax = {BLOCK: 8192, WAVE: 2048, GRID: 1 << 16}[engine]
    cases = [(W, m, "rand") for W, m in FP_EDGE]
    for W in [2, 4, 8, 16, 32, 64, 128, 512, 2048, 8192, 1 << 14, 1 << 16]:
        if W > wmax:
            continue
        for dens in (0.08, 0.3, 0.6, 0.92):
            cases.append((W, max(1, int(W * dens)), str(rng.choice(["rand", "left", "right"]))))
    for W, m, shape in cases:
        if W < 64 and engine == GRID:
            continue
        len_ = max(4 * W, 256)
        nwin = len_ // W
        widx = int(rng.integers(0, nwin))
        ws, we = widx * W + 1, (widx + 1) * W
        k, v, o, sems = random_partitioned_array(rng, len_, 0.5, max(1, len_ // 40))
        # re-place the window's content: m cells in the requested shape, ascending keys / a few semaphores with fresh ids
        inside = np.arange(ws - 1, we)
        o[inside] = 0
        if shape == "left":
            pos = inside[:m]
        elif shape == "right":
            pos = inside[W - m:]
        else:
            pos = np.sort(rng.choice(inside, size=m, replace=False))
        live = [int(s) for s in sems if not (ws <= s <= we)]
        sem_list = list(sems)
        for j, s in enumerate(pos):
            if j % 9 == 0 and m > 2:
                sem_list.append(int(s) + 1)
                k[s], v[s], o[s] = 0, float(len(sem_list)), 1
            else:
                k[s], v[s], o[s] = 5 + 2 * j, float(rng.integers(1, 1000)), 1
        sems2 = np.array([s if (s in live or i >= len(sems)) else 0 for i, s in enumerate(sem_list)], dtype=np.int64)
        # (table entries of the semaphores that used to sit in the window are tombstoned; the cells are gone)
        a = (k.copy(), v.copy(), o.copy(), sems2.copy())
        b = (k.copy(), v.copy(), o.copy(), sems2.copy())
        ref.rebalance(a[0], a[1], a[2], ws, we, a[3])
        dev.rebalance(b[0], b[1], b[2], ws, we, b[3])
        assert _same(a[:3], b[:3]), (W, m, shape)
        assert np.array_equal(a[3], b[3]), (W, m, shape)
        out = np.ones(len_, dtype=bool)
        out[ws - 1:we] = False
        assert np.array_equal(b[2][out], o[out]) and np.array_equal(b[0][out & o.astype(bool)], k[out & o.astype(bool)])
        assert int(b[2][ws - 1:we].sum()) == m


# ------------------------------------------------------------------ 3. restored layouts
def test_imported_vector_layout_behaves_like_the_original(dsa, hip, oracle):
    rng = np.random.default_rng(5)
    keys = np.sort(rng.choice(10 ** 6, size=20000, replace=False)) + 1
    vals = rng.random(20000) + 1.0
    src = dsa.dynamicsparsevec(keys, vals, binding=hip)
    more = rng.choice(10 ** 6, size=5000) + 1
    src.set_batch(more, np.where(rng.random(5000) < 0.2, 0.0, 2.0))
    k, v, o = src.export_layout()
    seg = src.info()["segment_capacity"]
    twins = [dsa.import_vector_layout(k, v, o, seg, n=len(src), binding=b) for b in (hip, oracle)]
    for t in twins:
        ti, si = t.info(), src.info()
        for f in ("capacity", "segment_capacity", "nb_segments", "nb_elements", "height"):
            assert ti[f] == si[f], f
        assert layouts_equal(t.export_layout(), (k, v, o))
    assert twins[0] == src
    assert twins[0].check()[2:7].sum() == 0
    ops_k = rng.choice(10 ** 6, size=30000) + 1
    ops_v = np.where(rng.random(30000) < 0.3, 0.0, 3.0)
    for t in twins + [src]:
        t.set_batch(ops_k, ops_v)
    assert layouts_equal(twins[0].export_layout(), twins[1].export_layout())
    assert layouts_equal(twins[0].export_layout(), src.export_layout())
    assert twins[0].info()["capacity"] == twins[1].info()["capacity"]


@pytest.mark.parametrize("shape", ["left", "right", "clumps"])
def test_root_rebalance_of_imported_skewed_layouts_matches_oracle(dsa, hip, oracle, shape):
    """layouts the reference only passes through in the middle of an operation (everything packed to one side, dense clumps
    between empty stretches), restored into both implementations and rebalanced at the root (src/pma.jl:94-103)."""
    rng = np.random.default_rng(9)
    cap, seg, m = 1 << 18, 16, 150000
    o = np.zeros(cap, dtype=np.uint8)
    if shape == "left":
        o[:m] = 1
    elif shape == "right":
        o[cap - m:] = 1
    else:
        starts = np.sort(rng.choice(cap // 4096, size=m // 3000, replace=False)) * 4096
        for s in starts:
            o[s:s + 3000] = 1
    n = int(o.sum())
    k = np.zeros(cap, dtype=np.int64)
    v = np.zeros(cap, dtype=np.float64)
    k[o.astype(bool)] = np.arange(1, n + 1) * 7
    v[o.astype(bool)] = rng.random(n) + 1.0
    a = dsa.import_vector_layout(k, v, o, seg, binding=hip)
    b = dsa.import_vector_layout(k, v, o, seg, binding=oracle)
    a.rebalance_root()
    b.rebalance_root()
    assert layouts_equal(a.export_layout(), b.export_layout())
    assert a.check()[2:7].sum() == 0
    kk, _, oo = a.export_layout()
    check_key_order(kk, oo)


def test_imported_packedcsc_layout_behaves_like_the_original(dsa, hip, oracle):
    rng = np.random.default_rng(11)
    rows = [np.sort(rng.choice(5000, size=int(rng.integers(0, 60)), replace=False)) + 1 for _ in range(300)]
    vals = [rng.random(len(r)) + 1.0 for r in rows]
    src = dsa.packedcsc(rows, vals, binding=hip)
    src.deletepartition(17)
    src[5, 300] = 2.5
    k, v, o, s = src.export_layout()
    seg = src.info()["segment_capacity"]
    twins = [dsa.import_packedcsc_layout(k, v, o, seg, s, binding=b) for b in (hip, oracle)]
    for t in twins:
        assert t.nbpartitions() == src.nbpartitions() and t.nnz() == src.nnz()
    for step in range(400):
        key, part = int(rng.integers(1, 5001)), int(rng.integers(1, 301))
        if part == 17:
            continue
        val = 0.0 if rng.random() < 0.3 else float(rng.integers(1, 100))
        for t in twins:
            t[key, part] = val
    la, lb = twins[0].export_layout(), twins[1].export_layout()
    assert layouts_equal(la[:3], lb[:3]) and np.array_equal(la[3], lb[3])
    check_semaphores(la[0], la[1], la[2], la[3])
