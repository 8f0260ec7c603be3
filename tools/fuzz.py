#!/usr/bin/env python3
"""Differential fuzzer (dev tool, GPU box): random write batches on matrices and vectors, HIP library vs the CPU oracle, slot
layout + tables + values compared after every batch.  Usage: python tools/fuzz.py [seconds] [first_seed]"""
import faulthandler
import os
import sys
import time

faulthandler.enable(all_threads=True)      # a SIGSEGV / SIGABRT in the native library prints the Python stack of every thread

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dsa_loader  # noqa: E402
from util import SplitMix64  # noqa: E402

dsa = dsa_loader.load()
BIG = os.environ.get("FUZZ_BIG") == "1"       # larger key spaces and batches: big-window yields, extends, > 1024 pending columns
hip = None if os.environ.get("FUZZ_SELF") == "1" else dsa.product()
sys.path.insert(0, os.path.join(ROOT, "oracle")); import oracle_binding; ora = oracle_binding.load(dsa)
if hip is None:
    hip = ora            # dry run of the script itself (oracle vs oracle)


def mat_equal(a, b, ctx):
    assert a.size() == b.size(), (ctx, a.size(), b.size())
    for o in (0, 1):
        La, Lb = a.export_layout(o), b.export_layout(o)
        for k in ("capacity", "segment_capacity", "nb_segments", "nb_elements", "height", "nb_partitions", "table_len"):
            assert La["info"][k] == Lb["info"][k], (ctx, o, k, La["info"][k], Lb["info"][k])
        for k in ("occ", "semaphores", "col_live"):
            assert np.array_equal(La[k], Lb[k]), (ctx, o, k)
        occ = La["occ"].astype(bool)
        assert np.array_equal(La["keys"][occ], Lb["keys"][occ]), (ctx, o, "keys")
        assert np.array_equal(La["vals"][occ], Lb["vals"][occ]), (ctx, o, "vals")
        live = La["col_live"].astype(bool)
        assert np.array_equal(La["col_keys"][live], Lb["col_keys"][live]), (ctx, o, "col_keys")


def mat_equal_after_error(a, b, ctx):
    """The state after a FAILED batch (a reference crash path): size(m) and the orientation that did not refuse the
    write are the reference's slot for slot (src/matrix.jl:43-62: colmajor holds the failing write when the rowmajor statement threw); the
    orientation that refused it keeps its tables where the reference leaves them half shifted — that one is not compared."""
    assert a.size() == b.size(), (ctx, a.size(), b.size())
    same = []
    for o in (0, 1):
        try:
            La, Lb = a.export_layout(o), b.export_layout(o)
            ok = all(La["info"][k] == Lb["info"][k] for k in ("capacity", "segment_capacity", "nb_segments", "nb_elements", "height", "nb_partitions", "table_len"))
            ok = ok and all(np.array_equal(La[k], Lb[k]) for k in ("occ", "semaphores", "col_live"))
            if ok:
                occ = La["occ"].astype(bool); live = La["col_live"].astype(bool)
                ok = (np.array_equal(La["keys"][occ], Lb["keys"][occ]) and np.array_equal(La["vals"][occ], Lb["vals"][occ])
                      and np.array_equal(La["col_keys"][live], Lb["col_keys"][live]))
        except Exception:
            ok = False
        same.append(bool(ok))
    assert any(same), (ctx, "neither orientation is the reference's after the failed batch", same)
    return same


def run_matrix(seed):
    g = SplitMix64(seed)
    spans = [3000, 40000, 300000, 2000000] if BIG else [50, 400, 3000, 40000]
    span_i = spans[g.next() % 4]
    span_j = spans[g.next() % 4]
    neg = g.next() % 4 == 0
    a = dsa.dynamicsparse(fill_mode=False, binding=hip)
    b = dsa.dynamicsparse(fill_mode=False, binding=ora)
    live = []
    steps = 4 + g.next() % 10
    for step in range(steps):
        nb = ([130, 2500, 9000, 30000, 60000, 120000] if BIG else [5, 40, 130, 600, 2500, 9000])[g.next() % 6]
        mode = g.next() % 5          # 0 random, 1 column stream (ascending), 2 row stream, 3 delete-heavy, 4 overwrite-heavy
        I, J, V = [], [], []
        if mode == 1 or mode == 2:
            base = (max([j for _, j in live], default=0) if mode == 1 else max([i for i, _ in live], default=0)) + 1
            per = 1 + g.next() % 20
            k = 0
            while len(I) < nb:
                rows = sorted({1 + int(g.next() % (span_i if mode == 1 else span_j)) for _ in range(per)})
                for r in rows:
                    if mode == 1:
                        I.append(r); J.append(base + k)
                    else:
                        I.append(base + k); J.append(r)
                    V.append(1.0 + (g.next() % 1000) / 1000.0)
                k += 1
        else:
            for _ in range(nb):
                r = g.next() % 10
                if live and ((mode == 3 and r < 6) or (mode != 3 and r < 1)):
                    i, j = live[g.next() % len(live)]
                    I.append(i); J.append(j); V.append(0.0)
                elif live and mode == 4 and r < 8:
                    i, j = live[g.next() % len(live)]
                    I.append(i); J.append(j); V.append(2.0 + (g.next() % 1000) / 1000.0)
                else:
                    i = 1 + int(g.next() % span_i); j = 1 + int(g.next() % span_j)
                    if neg and g.next() % 3 == 0:
                        i = -i
                    if neg and g.next() % 3 == 0:
                        j = -j
                    I.append(i); J.append(j); V.append(1.0 + (g.next() % 1000) / 1000.0)
        for (i, j, v) in zip(I, J, V):
            if v != 0.0:
                live.append((i, j))
        ea = eb = None
        try:
            a.set_batch(I, J, V)
        except dsa.DsaError as e:
            ea = e.code
        try:
            b.set_batch(I, J, V)
        except dsa.DsaError as e:
            eb = e.code
        assert ea == eb, ("error codes", seed, step, ea, eb)
        if ea is not None:
            mat_equal_after_error(a, b, (seed, step, mode, nb, "after error", ea))
            return "err%d" % ea
        mat_equal(a, b, (seed, step, mode, nb))
        if live:                              # views, lookups and the sparse-x product on the fresh state
            i0, j0 = live[g.next() % len(live)]
            for name, key in (("col_view", j0), ("row_view", i0)):
                assert getattr(a, name)(key) == getattr(b, name)(key), (seed, step, name, key)
            qi = [live[g.next() % len(live)][0] for _ in range(8)] + [1 + int(g.next() % span_i)]
            qj = [live[g.next() % len(live)][1] for _ in range(8)] + [1 + int(g.next() % span_j)]
            assert np.array_equal(a.get_batch(qi, qj), b.get_batch(qi, qj)), (seed, step, "get_batch")
            if not neg and g.next() % 3 == 0:      # (negative row keys: the reference's sparsevec throws — documented divergence)
                cols = sorted({j for _, j in live if j >= 1})
                xi = np.array(cols[:: max(1, len(cols) // (1 + g.next() % 40))], dtype=np.int64)
                xv = 1.0 + (np.arange(len(xi)) % 5) / 4.0
                ia, va2 = a.mul((xi, xv)); ib, vb2 = b.mul((xi, xv))
                assert np.array_equal(ia, ib) and np.allclose(va2, vb2, rtol=1e-12, atol=0), (seed, step, "sparse-x mul")
        if g.next() % 7 == 0 and live:       # tombstones now and then
            i, j = live[g.next() % len(live)]
            try:
                a.deletecolumn(j); b.deletecolumn(j)
            except dsa.DsaError:
                pass
            live = [(p, q) for (p, q) in live if q != j]
            mat_equal(a, b, (seed, step, "deletecolumn"))
    m, n = a.size()
    if n >= 1:
        x = 1.0 + np.arange(n) % 7 / 8.0
        ya, yb = a.mul(x), b.mul(x)
        assert np.allclose(ya, yb, rtol=1e-12, atol=0), (seed, "spmv")
    if m >= 1 and n >= 1:                  # transpose(mat) * v walks the other orientation with the same kernel
        xt = 1.0 + np.arange(m) % 5 / 4.0
        ya, yb = dsa.Transposed(a).mul(xt), dsa.Transposed(b).mul(xt)
        assert np.allclose(ya, yb, rtol=1e-12, atol=0), (seed, "spmv transposed")
    return "ok"


def run_vector(seed):
    g = SplitMix64(seed)
    span = [100, 5000, 10**6][g.next() % 3]
    a = dsa.dynamicsparsevec([], [], binding=hip)
    b = dsa.dynamicsparsevec([], [], binding=ora)
    for step in range(3 + g.next() % 8):
        nb = [10, 200, 3000, 20000][g.next() % 4]
        mode = g.next() % 3
        if BIG and mode == 1:
            nb = [20000, 60000, 150000][g.next() % 3]      # long ascending runs: the count model of the append replay, with extends
        if mode == 0:
            keys = 1 + (np.array([g.next() for _ in range(nb)], dtype=np.uint64) % np.uint64(span)).astype(np.int64)
        elif mode == 1:
            base = int(a.export_layout()[0].max()) + 1 if a.nnz() else 1
            keys = np.arange(base, base + nb, dtype=np.int64)
        else:
            keys = -(1 + (np.array([g.next() for _ in range(nb)], dtype=np.uint64) % np.uint64(span)).astype(np.int64))
        vals = np.where(np.array([g.next() % 6 for _ in range(nb)]) == 0, 0.0, 1.5)
        a.set_batch(keys, vals); b.set_batch(keys, vals)
        ka, kb = a.export_layout(), b.export_layout()
        assert np.array_equal(ka[2], kb[2]), (seed, step, "occ")
        occ = ka[2].astype(bool)
        assert np.array_equal(ka[0][occ], kb[0][occ]) and np.array_equal(ka[1][occ], kb[1][occ]), (seed, step, "cells")
        assert a.info() ["capacity"] == b.info()["capacity"]
        if step % 2 == 1:                     # whole-vector operations against a second vector built another way
            k2, v2 = b.nonzeros()
            sel = np.array([g.next() % 4 != 0 for _ in range(len(k2))], dtype=bool)
            bump = np.where(np.array([g.next() % 5 == 0 for _ in range(len(k2))]), -1.0, 1.0)
            a2 = dsa.dynamicsparsevec(k2[sel], (v2 * bump)[sel], binding=hip)
            b2 = dsa.dynamicsparsevec(k2[sel], (v2 * bump)[sel], binding=ora)
            for x, y in ((a + a2, b + b2), (a - a2, b - b2), (a2 - a, b2 - b), (a.axpby(0.5, a2, 2.0), b.axpby(0.5, b2, 2.0))):
                assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]), (seed, step, "axpby")
            assert (a == a2) == (b == b2) and (a == a) and ((a2 == a) == (b2 == b)), (seed, step, "==")
            a3 = dsa.dynamicsparsevec(*a.nonzeros(), n=len(a), binding=hip)
            assert a3 == a and a == a3, (seed, step, "== rebuilt")
    return "ok"


def run_shared_words(seed):
    """Targeted stress of the batch-parallel rounds (DESIGN §3.2b invariant): many writes of ONE batch land in the same
    64-slot occupancy words — small arrays, dense key ranges — so that k_apply waves on different XCDs update bits of the
    same word in the same launch (device-scope atomics) while their slot footprints stay disjoint."""
    g = SplitMix64(seed)
    n0 = [300, 2000, 20000][g.next() % 3]
    stride = 2 + g.next() % 6
    keys0 = np.arange(1, n0 + 1, dtype=np.int64) * stride
    a = dsa.dynamicsparsevec(keys0, np.ones(n0), binding=hip)
    b = dsa.dynamicsparsevec(keys0, np.ones(n0), binding=ora)
    for step in range(6):
        nb = [200, 1000, 4000][g.next() % 3]
        raw = np.array([g.next() for _ in range(nb)], dtype=np.uint64)
        keys = 1 + (raw % np.uint64(n0 * stride)).astype(np.int64)
        vals = np.where((raw >> np.uint64(40)) % np.uint64(5) == 0, 0.0, 2.5)      # 20 % deletes
        a.set_batch(keys, vals); b.set_batch(keys, vals)
        ka, kb = a.export_layout(), b.export_layout()
        assert np.array_equal(ka[2], kb[2]), (seed, step, "occ")
        occ = ka[2].astype(bool)
        assert np.array_equal(ka[0][occ], kb[0][occ]) and np.array_equal(ka[1][occ], kb[1][occ]), (seed, step, "cells")
        if hip is not ora:
            assert not a.check()[2:7].any(), (seed, step, "invariant checker")
    return "ok"


def run_same_leaf(seed):
    """Targeted stress of the resolver's count bookkeeping (csrc/parbatch.hip, tight footprints): several inserts / deletes of ONE batch in
    the SAME leaf — chosen from the oracle's layout, half of them in leaves that end on a multiple of 4096 slots (the cells of the resolver's
    spatial hash; a leaf's last slot is the first slot of the next cell) — spread over a batch of far-away value updates so that they meet
    in one round.  Vectors above 2^16 slots (grid rounds, 16-slot segments) and, for 8-slot segments at that size, grown matrices are covered
    by run_matrix; here the geometry is the built one."""
    g = SplitMix64(seed)
    n0 = [3000, 12000, 50000, 60007, 100000, 180000, 300000][g.next() % 7]      # (the two small ones: the one-workgroup rounds)
    stride = 4 + 2 * (g.next() % 3)
    keys0 = np.arange(1, n0 + 1, dtype=np.int64) * stride
    a = dsa.dynamicsparsevec(keys0, np.ones(n0), binding=hip)
    b = dsa.dynamicsparsevec(keys0, np.ones(n0), binding=ora)
    for step in range(5):
        K, V, O = b.export_layout()
        inf = b.info()
        cap, seg = inf["capacity"], inf["segment_capacity"]
        occ = O.astype(bool)
        ops = []                                   # (key, value) of the leaf ops, in the order they must keep
        nleaves = 6 + g.next() % 20
        for _ in range(nleaves):
            pick = g.next() % 8
            if pick < 3 and cap >= 8192:
                kcell = 1 + g.next() % max(1, cap // 4096 - 1)
                last = 4096 * kcell + seg * (int(g.next() % 3) - 1)      # the boundary leaf or a neighbour (1-based last slot)
            elif pick == 3:
                last = cap - seg * int(g.next() % 3)                     # the end of the array (left-falling inserts)
            elif pick == 4:
                last = seg * (1 + int(g.next() % 3))                     # ... and its start
            else:
                last = seg * (1 + g.next() % (cap // seg))
            lo0, hi0 = last - seg, last                                   # 0-based slice of the leaf
            cells = [int(x) for x in K[lo0:hi0][occ[lo0:hi0]]]
            if not cells:
                continue
            nops = 2 + g.next() % 4
            for _ in range(nops):
                r = g.next() % 10
                if r < 6 and cells:                                      # delete (the last cell of the leaf more often than not)
                    c = cells.pop(-1 if g.next() % 2 else int(g.next() % len(cells)))
                    ops.append((c, 0.0))
                else:                                                     # insert between / behind the cells of the leaf
                    base = cells[int(g.next() % len(cells))] if cells else int(K[lo0:hi0].max())
                    ops.append((base + 1 + int(g.next() % (stride - 1)), 3.25))
        # far-away fillers: value updates of existing keys (no footprint in common with anything), a few hundred around the leaf ops
        nfill = [150, 400, 900][g.next() % 3]
        pool_keys = K[occ]
        fill = pool_keys[(np.array([g.next() for _ in range(nfill)], dtype=np.uint64) % np.uint64(len(pool_keys))).astype(np.int64)]
        touched = {k for k, _ in ops}
        fill = [int(k) for k in fill if int(k) not in touched]
        keys, vals = [], []
        slots = sorted(int(g.next() % (len(fill) + 1)) for _ in ops)
        it = iter(zip(slots, ops)); nxt = next(it, None)
        for idx, fk in enumerate(fill + [None]):
            while nxt is not None and nxt[0] == idx:
                keys.append(nxt[1][0]); vals.append(nxt[1][1]); nxt = next(it, None)
            if fk is not None:
                keys.append(fk); vals.append(1.0 + (idx % 7) / 8.0)
        keys = np.array(keys, dtype=np.int64); vals = np.array(vals)
        a.set_batch(keys, vals); b.set_batch(keys, vals)
        ka, kb = a.export_layout(), b.export_layout()
        assert a.info()["capacity"] == b.info()["capacity"], (seed, step, "capacity")
        assert np.array_equal(ka[2], kb[2]), (seed, step, "occ")
        o2 = ka[2].astype(bool)
        assert np.array_equal(ka[0][o2], kb[0][o2]) and np.array_equal(ka[1][o2], kb[1][o2]), (seed, step, "cells")
        ia, ib = a.info(), b.info()
        assert ia["stat_rebalances"] == ib["stat_rebalances"] and ia["stat_window_slots"] == ib["stat_window_slots"], (seed, step, "statistics")
    return "ok"


def run_same_leaf_matrix(seed):
    """run_same_leaf for a MATRIX: element inserts / deletes, NEW columns (their semaphore lands in front of an existing column's) and
    writes that empty a column, several per leaf of the colmajor orientation and per batch, leaves picked from the oracle's layout (half of
    them ending on a multiple of 4096 slots), between far-away overwrites.  The rowmajor twin receives the transposed ops."""
    g = SplitMix64(seed)
    m = [3000, 40000][g.next() % 2]
    ncols = [12000, 30000, 70000][g.next() % 3]
    per = 1 + g.next() % 4
    cols = np.repeat(np.arange(1, ncols + 1, dtype=np.int64) * 4, per)
    rows = 2 + 2 * (np.array([g.next() for _ in range(len(cols))], dtype=np.uint64) % np.uint64(m // 2)).astype(np.int64)      # even rows: odd ones are free
    a = dsa.dynamicsparse(rows, cols, np.ones(len(cols)), binding=hip)
    b = dsa.dynamicsparse(rows, cols, np.ones(len(cols)), binding=ora)
    mat_equal(a, b, (seed, "build"))
    colset = set(int(c) for c in cols)
    for step in range(4):
        L = b.export_layout(0)
        K, Vv, O = L["keys"], L["vals"], L["occ"].astype(bool)
        cap, seg = L["info"]["capacity"], L["info"]["segment_capacity"]
        sem_pos = L["semaphores"]; ck = L["col_keys"]; live = L["col_live"].astype(bool)
        order = np.argsort(np.where(live, sem_pos, np.iinfo(np.int64).max))
        spos_sorted = sem_pos[order][: int(live.sum())]            # 1-based slots of the live semaphores, ascending
        def column_of_slot(slot1):                                  # column key of the partition that holds 1-based slot `slot1`
            q = int(np.searchsorted(spos_sorted, slot1, side="right")) - 1
            return int(ck[order[q]]) if q >= 0 else None
        ops = []
        for _ in range(6 + g.next() % 16):
            pick = g.next() % 8
            if pick < 3:
                kcell = 1 + g.next() % max(1, cap // 4096 - 1)
                last = 4096 * kcell + seg * (int(g.next() % 3) - 1)
            elif pick == 3:
                last = cap - seg * int(g.next() % 3)
            elif pick == 4:
                last = seg * (1 + int(g.next() % 3))
            else:
                last = seg * (1 + g.next() % (cap // seg))
            lo0, hi0 = last - seg, last
            slots = [lo0 + int(x) for x in np.nonzero(O[lo0:hi0])[0]]
            if not slots:
                continue
            for _ in range(2 + g.next() % 4):
                s0 = slots[int(g.next() % len(slots))] if g.next() % 3 else slots[-1]
                col = column_of_slot(s0 + 1)
                if col is None:
                    continue
                r = g.next() % 10
                if K[s0] == 0:                                            # a semaphore cell
                    if r < 5:
                        newc = col - 1 - int(g.next() % 3)
                        if newc >= 1 and newc not in colset:
                            colset.add(newc); ops.append((1 + 2 * int(g.next() % (m // 2)), newc, 2.5)); continue
                    ops.append((1 + 2 * int(g.next() % 8), col, 2.75))    # a small odd row: right behind the semaphore
                elif r < 5:
                    ops.append((int(K[s0]), col, 0.0))                    # delete the element
                else:
                    ops.append((max(1, int(K[s0]) + (1 if g.next() % 2 else -1)), col, 3.25))      # a row next to it (never 0: the semaphore key)
        nfill = [150, 400, 900][g.next() % 3]
        pick = (np.array([g.next() for _ in range(nfill)], dtype=np.uint64) % np.uint64(len(cols))).astype(np.int64)
        touched = {(i, j) for i, j, _ in ops}
        fill = [(int(rows[t]), int(cols[t])) for t in pick if (int(rows[t]), int(cols[t])) not in touched]
        I, J, V = [], [], []
        where = sorted(int(g.next() % (len(fill) + 1)) for _ in ops)
        it = iter(zip(where, ops)); nxt = next(it, None)
        for idx, f in enumerate(fill + [None]):
            while nxt is not None and nxt[0] == idx:
                I.append(nxt[1][0]); J.append(nxt[1][1]); V.append(nxt[1][2]); nxt = next(it, None)
            if f is not None:
                I.append(f[0]); J.append(f[1]); V.append(1.0 + (idx % 7) / 8.0)
        # (a filler may address an element an earlier leaf op deleted: it then re-inserts it — same on both sides)
        a.set_batch(I, J, V); b.set_batch(I, J, V)
        mat_equal(a, b, (seed, step, "same leaf matrix", len(I)))
        for o in (0, 1):
            ia, ib = a.info(o), b.info(o)
            assert ia["stat_rebalances"] == ib["stat_rebalances"] and ia["stat_window_slots"] == ib["stat_window_slots"], (seed, step, o, "statistics")
    return "ok"


def _both(fa, fb):
    """runs the same call on both libraries; returns the (equal) error code or None"""
    ea = eb = None
    try:
        fa()
    except dsa.DsaError as e:
        ea = e.code
    try:
        fb()
    except dsa.DsaError as e:
        eb = e.code
    assert ea == eb, ("error codes", ea, eb)
    return ea


def run_tombstones(seed):
    """Targeted stress of the partition tables: deletecolumn! / deleterow! leave tombstones (src/pcsr.jl:188-212, src/matrix.jl:95-111), then
    batches create columns NEXT to tombstoned ids, re-create deleted ones (addpartition!(pcsc, prev) reuses a tombstoned id or shifts the
    tables, src/pcsr.jl:114-146) and write into the survivors; both orientations compared after every call."""
    g = SplitMix64(seed)
    m = [400, 5000][g.next() % 2]
    ncols = [300, 3000, 20000][g.next() % 3]
    per = 1 + g.next() % 3
    cols = np.repeat(np.arange(1, ncols + 1, dtype=np.int64) * 3, per)
    rows = 1 + (np.array([g.next() for _ in range(len(cols))], dtype=np.uint64) % np.uint64(m)).astype(np.int64)
    a = dsa.dynamicsparse(rows, cols, np.ones(len(cols)), binding=hip)
    b = dsa.dynamicsparse(rows, cols, np.ones(len(cols)), binding=ora)
    mat_equal(a, b, (seed, "build"))
    alive = sorted(set(int(c) for c in cols)); dead = []
    top_id = 3 * ncols
    for step in range(6):
        for _ in range(1 + g.next() % 6):                      # tombstones
            if not alive:
                break
            j = alive.pop(int(g.next() % len(alive))); dead.append(j)
            if _both(lambda: a.deletecolumn(j), lambda: b.deletecolumn(j)) is not None:
                return "err"
            mat_equal(a, b, (seed, step, "deletecolumn", j))
        if g.next() % 3 == 0:
            i = 1 + int(g.next() % m)
            if _both(lambda: a.deleterow(i), lambda: b.deleterow(i)) is not None:
                return "err"
            mat_equal(a, b, (seed, step, "deleterow", i))
        I, J, V = [], [], []
        for _ in range([3, 40, 300, 2000][g.next() % 4]):
            # (a column created BETWEEN two live ones while a tombstone sits further up the table runs into the reference's @assert,
            # src/pcsr.jl:132 — a crash path, the scenario would end there: every tombstone is used once, by its own id or a neighbour,
            # which lands ON it; new ids beyond the last column take the push branch)
            r = g.next() % 10
            v = 0.0 if g.next() % 6 == 0 else 1.5
            if r < 4 and dead and v != 0.0:
                d = dead.pop(int(g.next() % len(dead)))
                j = max(1, d + int(g.next() % 3) - 1)
                alive.append(j)
            elif r == 4 and v != 0.0:
                top_id += 1 + int(g.next() % 3); j = top_id; alive.append(j)
            elif alive:
                j = alive[int(g.next() % len(alive))]
            else:
                continue
            I.append(1 + int(g.next() % m)); J.append(j); V.append(v)
        e = _both(lambda: a.set_batch(I, J, V), lambda: b.set_batch(I, J, V))
        if e is not None:
            mat_equal_after_error(a, b, (seed, step, "after error", e))
            return "err%d" % e
        mat_equal(a, b, (seed, step, "batch", len(I)))
        alive = sorted(set(alive))
    n = a.size()[1]
    if n >= 1:
        x = 1.0 + np.arange(n) % 7 / 8.0
        assert np.allclose(a.mul(x), b.mul(x), rtol=1e-12, atol=0), (seed, "spmv")
    return "ok"


def run_crash_paths(seed):
    """The reference's crash paths on purpose: a sparse key space with tombstones in BOTH tables, batches whose keys
    fall between live ones / behind a tombstoned tail, so that most scenarios end in an AssertionError or BoundsError of addpartition!
    somewhere inside a batch.  Status, size(m) and the orientation that did not refuse the write must be the reference's
    (src/matrix.jl:43-62: colmajor, then rowmajor, write by write)."""
    g = SplitMix64(seed)
    nk = [20, 60, 400][g.next() % 3]
    keys = np.arange(1, nk + 1, dtype=np.int64) * 3
    per = 1 + g.next() % 4
    I0 = np.repeat(keys, per)
    J0 = keys[(np.array([g.next() for _ in range(len(I0))], dtype=np.uint64) % np.uint64(nk)).astype(np.int64)]
    a = dsa.dynamicsparse(I0, J0, np.ones(len(I0)), binding=hip)
    b = dsa.dynamicsparse(I0, J0, np.ones(len(I0)), binding=ora)
    for step in range(5):
        for _ in range(g.next() % 4):
            k = int(keys[g.next() % nk])
            which = g.next() % 2
            f = (lambda: (a.deleterow(k) if which else a.deletecolumn(k)), lambda: (b.deleterow(k) if which else b.deletecolumn(k)))
            e = _both(*f)
            if e is not None and e != dsa.binding.EARG:          # (EARG: the column / row was deleted before — nothing happened)
                mat_equal_after_error(a, b, (seed, step, "delete", k)); return "err%d" % e
            if e is None:
                mat_equal(a, b, (seed, step, "delete", k))
        nb = [2, 8, 40, 200, 700][g.next() % 5]
        top = 3 * nk
        I, J, V = [], [], []
        for _ in range(nb):
            r = g.next() % 16
            i = int(keys[g.next() % nk]); j = int(keys[g.next() % nk])
            if r == 0: i = 1 + int(g.next() % (3 * nk + 6))            # any row key: between live ones, on a tombstone, behind the tail
            if r == 1: j = 1 + int(g.next() % (3 * nk + 6))
            if r == 2: top += 3; i = top
            if r == 3: top += 3; j = top
            I.append(i); J.append(j); V.append(0.0 if g.next() % 7 == 0 else 1.0 + (g.next() % 8))
        e = _both(lambda: a.set_batch(I, J, V), lambda: b.set_batch(I, J, V))
        if e is not None:
            same = mat_equal_after_error(a, b, (seed, step, "batch", nb))
            return "err%d%s" % (e, "c" if not same[0] else ("r" if not same[1] else "x"))
        mat_equal(a, b, (seed, step, "batch", nb))
    return "ok"


def run_packedcsc(seed):
    """The PackedCSC API on its own (explicit partition ids, src/pcsr.jl:26-63, 171-232, 294-339): writes, deletes, trailing partitions
    created by a write behind the last one, deletepartition! and the error paths behind it, lookups; layout compared after every step."""
    g = SplitMix64(seed)
    nparts = [3, 40, 600][g.next() % 3]
    span = [30, 500, 20000][g.next() % 3]
    rk, vv = [], []
    for _ in range(nparts):
        ks = sorted({1 + int(g.next() % span) for _ in range(int(g.next() % 12))})
        rk.append(ks); vv.append([1.0 + (k % 9) / 8.0 for k in ks])
    a = dsa.packedcsc(rk, vv, binding=hip)
    b = dsa.packedcsc(rk, vv, binding=ora)
    def same():
        La, Lb = a.export_layout(), b.export_layout()
        assert a.info()["capacity"] == b.info()["capacity"] and a.nbpartitions() == b.nbpartitions() and a.nnz() == b.nnz()
        assert np.array_equal(La[2], Lb[2]), (seed, "occ")
        o = La[2].astype(bool)
        assert np.array_equal(La[0][o], Lb[0][o]) and np.array_equal(La[1][o], Lb[1][o]) and np.array_equal(La[3], Lb[3]), (seed, "cells / semaphores")
    same()
    top = nparts
    for step in range(30 + int(g.next() % 200)):
        r = g.next() % 20
        if r == 0 and top >= 1:
            p = 1 + int(g.next() % (top + 1))
            e = _both(lambda: a.deletepartition(p), lambda: b.deletepartition(p))
        else:
            p = 1 + int(g.next() % (top + (2 if r < 3 else 0)))            # now and then a write behind the last partition: trailing partitions
            k = 1 + int(g.next() % span)
            v = 0.0 if g.next() % 5 == 0 else 2.0 + (step % 5) / 4.0
            def wa(): a[k, p] = v
            def wb(): b[k, p] = v
            e = _both(wa, wb)
            if e is None and v != 0.0:
                top = max(top, p)
        if e is not None and e not in (dsa.binding.EDELETED, dsa.binding.EBOUNDS):
            return "err%d" % e
        same()
        if step % 7 == 0:
            p = 1 + int(g.next() % max(top, 1)); k = 1 + int(g.next() % span)
            ga = gb = None
            try: ga = a[k, p]
            except dsa.DsaError as e2: ga = ("err", e2.code)
            try: gb = b[k, p]
            except dsa.DsaError as e2: gb = ("err", e2.code)
            assert ga == gb, (seed, step, "get", ga, gb)
    return "ok"


def run_fill(seed):
    """Fill mode (src/buffer.jl, src/matrix.jl:113-134): addrow! (a second addrow! of a row is rejected), element appends through setindex!
    (duplicates of (i, j) are accumulated with +: integer-valued values, the fold order is the reference's unstable sort), closefillmode!,
    then ordinary writes and slices / views on the flushed matrix."""
    g = SplitMix64(seed)
    m = [50, 2000, 60000][g.next() % 3]
    n = [40, 3000, 50000][g.next() % 3]
    a = dsa.dynamicsparse(fill_mode=True, binding=hip)
    b = dsa.dynamicsparse(fill_mode=True, binding=ora)
    for _ in range(int(g.next() % 400)):
        r = 1 + int(g.next() % m)
        k = int(g.next() % 12)
        cs = sorted({1 + int(g.next() % n) for _ in range(k)})
        vs = [float(1 + g.next() % 5) for _ in cs]
        e = _both(lambda: a.addrow(r, cs, vs), lambda: b.addrow(r, cs, vs))
        if e not in (None, dsa.binding.EMODE):          # EMODE: "Row already written in dynamic sparse matrix buffer." (src/buffer.jl:13) — the call is rejected, the buffer unchanged
            return "err%d addrow" % e
    ne = [0, 50, 3000, 40000][g.next() % 4]
    if ne:
        I = 1 + (np.array([g.next() for _ in range(ne)], dtype=np.uint64) % np.uint64(m)).astype(np.int64)
        J = 1 + (np.array([g.next() for _ in range(ne)], dtype=np.uint64) % np.uint64(n)).astype(np.int64)
        V = (1 + np.array([g.next() % 4 for _ in range(ne)])).astype(np.float64)
        e = _both(lambda: a.set_batch(I, J, V), lambda: b.set_batch(I, J, V))
        if e is not None:
            return "err%d appends" % e
    e = _both(lambda: a.closefillmode(), lambda: b.closefillmode())
    if e is not None:
        return "err%d closefillmode" % e
    mat_equal(a, b, (seed, "closefillmode"))
    assert a.nnz() == b.nnz()
    for step in range(3):
        nb = [5, 200, 4000][g.next() % 3]
        I = 1 + (np.array([g.next() for _ in range(nb)], dtype=np.uint64) % np.uint64(m)).astype(np.int64)
        J = 1 + (np.array([g.next() for _ in range(nb)], dtype=np.uint64) % np.uint64(n)).astype(np.int64)
        V = np.where(np.array([g.next() % 4 for _ in range(nb)]) == 0, 0.0, 2.5)
        e = _both(lambda: a.set_batch(I, J, V), lambda: b.set_batch(I, J, V))
        if e is not None:
            return "err%d" % e
        mat_equal(a, b, (seed, step, "writes behind the flush"))
        j0, i0 = int(J[0]), int(I[0])
        assert a.col_view(j0) == b.col_view(j0) and a.row_view(i0) == b.row_view(i0), (seed, step, "views")
    return "ok"


def run_shrink_grow(seed):
    """_shrink! and _extend! back to back (src/pma.jl:135-161): a vector or a matrix is emptied in batches — ascending, descending or random
    key order — down to a handful of cells (several halvings), then filled again past its old size; layouts compared after every batch."""
    g = SplitMix64(seed)
    n0 = [3000, 40000, 250000 if BIG else 90000][g.next() % 3]
    keys = np.arange(1, n0 + 1, dtype=np.int64) * 3
    order = g.next() % 3
    if g.next() % 2 == 0:
        a = dsa.dynamicsparsevec(keys, np.ones(n0), binding=hip); b = dsa.dynamicsparsevec(keys, np.ones(n0), binding=ora)
        def same(ctx):
            ka, kb = a.export_layout(), b.export_layout()
            assert a.info()["capacity"] == b.info()["capacity"], (ctx, "capacity", a.info()["capacity"], b.info()["capacity"])
            assert np.array_equal(ka[2], kb[2]), (ctx, "occ")
            o = ka[2].astype(bool)
            assert np.array_equal(ka[0][o], kb[0][o]) and np.array_equal(ka[1][o], kb[1][o]), (ctx, "cells")
        def write(ks, vs):
            a.set_batch(ks, vs); b.set_batch(ks, vs)
    else:
        per = 1 + g.next() % 3
        cols = np.repeat(np.arange(1, n0 // per + 1, dtype=np.int64), per)
        rows = 1 + (np.array([g.next() for _ in range(len(cols))], dtype=np.uint64) % np.uint64(5000)).astype(np.int64)
        a = dsa.dynamicsparse(rows, cols, np.ones(len(cols)), binding=hip); b = dsa.dynamicsparse(rows, cols, np.ones(len(cols)), binding=ora)
        keys = np.arange(len(cols), dtype=np.int64)                 # indices of the triples
        def same(ctx):
            mat_equal(a, b, ctx)
        def write(ks, vs):
            a.set_batch(rows[ks], cols[ks], vs); b.set_batch(rows[ks], cols[ks], vs)
    same((seed, "build"))
    perm = keys.copy()
    if order == 1:
        perm = perm[::-1].copy()
    elif order == 2:
        rnd = np.array([g.next() for _ in range(len(perm))], dtype=np.uint64)
        perm = perm[np.argsort(rnd, kind="stable")]
    keep = 3 + int(g.next() % 40)
    pos = 0
    while pos < len(perm) - keep:
        nb = min(len(perm) - keep - pos, [200, 3000, 30000][g.next() % 3])
        write(perm[pos:pos + nb], np.zeros(nb)); pos += nb
        same((seed, "emptying", pos))
    pos = 0
    while pos < len(perm):
        nb = min(len(perm) - pos, [500, 8000, 60000][g.next() % 3])
        write(perm[pos:pos + nb], np.full(nb, 2.0)); pos += nb
        same((seed, "refilling", pos))
    return "ok"


def run_hot_keys(seed):
    """A few HOT keys written over and over inside one batch — insert, delete, overwrite of the same key and of its direct neighbours, in
    every order — between ordinary writes: identical and adjacent footprints, prefixes of one or two ops, hand-overs between the rounds
    and the sequencer in both directions.  Sequential semantics: the last write of a key wins, every intermediate state decides the
    rebalances in between."""
    g = SplitMix64(seed)
    n0 = [200, 5000, 70000][g.next() % 3]
    stride = 5
    keys0 = np.arange(1, n0 + 1, dtype=np.int64) * stride
    if g.next() % 2 == 0:
        a = dsa.dynamicsparsevec(keys0, np.ones(n0), binding=hip); b = dsa.dynamicsparsevec(keys0, np.ones(n0), binding=ora)
        def write(ks, vs):
            a.set_batch(ks, vs); b.set_batch(ks, vs)
        def same(ctx):
            ka, kb = a.export_layout(), b.export_layout()
            assert a.info()["capacity"] == b.info()["capacity"], (ctx, "capacity")
            assert np.array_equal(ka[2], kb[2]), (ctx, "occ")
            o = ka[2].astype(bool)
            assert np.array_equal(ka[0][o], kb[0][o]) and np.array_equal(ka[1][o], kb[1][o]), (ctx, "cells")
            assert a.info()["stat_rebalances"] == b.info()["stat_rebalances"], (ctx, "rebalances")
    else:
        ncol = max(4, n0 // 6)
        cols0 = 1 + (np.arange(n0) % ncol).astype(np.int64) * 2
        a = dsa.dynamicsparse(keys0, cols0, np.ones(n0), binding=hip); b = dsa.dynamicsparse(keys0, cols0, np.ones(n0), binding=ora)
        hot_col = [int(c) for c in cols0[:8]]
        def write(ks, vs):
            cs = np.array([hot_col[int(k) % 8] for k in ks], dtype=np.int64)      # a key always goes to the same column
            a.set_batch(ks, cs, vs); b.set_batch(ks, cs, vs)
        def same(ctx):
            mat_equal(a, b, ctx)
    for step in range(5):
        nhot = 1 + int(g.next() % 6)
        hot = [int(keys0[int(g.next() % n0)]) + int(g.next() % 3) - 1 for _ in range(nhot)]
        hot = [h for h in hot if h >= 1] or [3]
        nb = [20, 300, 3000][g.next() % 3]
        ks, vs = [], []
        for _ in range(nb):
            r = g.next() % 10
            if r < 7:
                h = hot[int(g.next() % len(hot))] + (int(g.next() % 3) - 1 if g.next() % 4 == 0 else 0)
                ks.append(max(1, h)); vs.append(0.0 if g.next() % 2 else 1.0 + (g.next() % 8) / 8.0)
            else:
                ks.append(1 + int(g.next() % (n0 * stride))); vs.append(0.0 if g.next() % 5 == 0 else 4.5)
        write(np.array(ks, dtype=np.int64), np.array(vs))
        same((seed, step, "hot keys", nb))
    return "ok"


def run_append_models(seed):
    """Targeted stress of the count-only append replay (csrc/appendmodel.hip): structures BUILT from data (16-slot segments: the
    geometry on which typed runs — semaphore cells of new columns — are count-only too) or grown from a few keys (small segments),
    then append runs of random lengths around the model's minimum (512), across extends, interleaved with random writes, deletes
    and column deletions that leave tombstones and ragged tails; layouts, tables and rebalance statistics after every batch."""
    g = SplitMix64(seed)
    if g.next() % 2 == 0:
        n0 = [3, 200, 50000, 90000, 400000][g.next() % 5]
        grown = g.next() % 2 == 0 and n0 >= 50000
        keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 3
        first = 3 if grown else n0
        a = dsa.dynamicsparsevec(keys0[:first], np.ones(first), binding=hip)
        b = dsa.dynamicsparsevec(keys0[:first], np.ones(first), binding=ora)
        if grown:
            a.set_batch(keys0[first:], np.ones(n0 - first)); b.set_batch(keys0[first:], np.ones(n0 - first))
        top = int(keys0[-1])
        for step in range(3 + g.next() % 5):
            r = [300, 511, 512, 513, 3000, 40000, 150000][g.next() % 7]
            ks = top + np.cumsum(1 + (np.array([g.next() for _ in range(r)], dtype=np.uint64) % np.uint64(3)).astype(np.int64))
            top = int(ks[-1])
            pre = np.array([keys0[g.next() % n0], top - 1], dtype=np.int64)          # a breaker in front, an update behind
            kk = np.concatenate([pre[:1], ks, pre[1:]]); vv = np.concatenate([[0.0 if g.next() % 2 else 2.5], np.full(r, 1.5), [3.5]])
            a.set_batch(kk, vv); b.set_batch(kk, vv)
            ka, kb = a.export_layout(), b.export_layout()
            assert np.array_equal(ka[2], kb[2]), (seed, step, "occ")
            occ = ka[2].astype(bool)
            assert np.array_equal(ka[0][occ], kb[0][occ]) and np.array_equal(ka[1][occ], kb[1][occ]), (seed, step, "cells")
            ia, ib = a.info(), b.info()
            for k in ("capacity", "stat_extends", "stat_rebalances", "stat_window_slots"):
                assert ia[k] == ib[k], (seed, step, k, ia[k], ib[k])
        return "ok"
    m = [2000, 40000][g.next() % 2]
    n0 = [3000, 12000][g.next() % 2]
    grown_m = g.next() % 2 == 0
    if grown_m:
        # a matrix GROWN from the empty one keeps 8-slot segments for life: its typed runs take the per-epoch replay (k_append_model5:
        # cross-leaf shifts of semaphores, surviving-event reconstruction, trailing epochs handed to the per-op replay)
        n0 = [1, 40, 700, 4000, 9000][g.next() % 5]
    I0, J0 = [], []
    for j in range(1, n0 + 1):
        for i in sorted({1 + int(g.next() % m) for _ in range(1 + g.next() % 10)}):
            I0.append(i); J0.append(j)
    V0 = 1.0 + np.arange(len(I0)) % 9 / 8.0
    if grown_m:
        a = dsa.dynamicsparse(fill_mode=False, binding=hip)
        b = dsa.dynamicsparse(fill_mode=False, binding=ora)
        a.set_batch(I0, J0, V0); b.set_batch(I0, J0, V0)
        mat_equal(a, b, (seed, "grown"))
    else:
        a = dsa.dynamicsparse(I0, J0, V0, binding=hip)
        b = dsa.dynamicsparse(I0, J0, V0, binding=ora)
    col = n0
    for step in range(3 + g.next() % 4):
        ncols = [40, 60, 700, 5000][g.next() % 4]
        maxlen = [1, 8, 40][g.next() % 3]
        I, J = [], []
        for _ in range(ncols):
            col += 1
            for i in sorted({1 + int(g.next() % m) for _ in range(1 + g.next() % maxlen)}):
                I.append(i); J.append(col)
        V = 1.0 + np.arange(len(I)) % 7 / 8.0
        if g.next() % 3 == 0:                 # a write to an old column in front: the run starts behind it
            I.insert(0, 1 + int(g.next() % m)); J.insert(0, 1 + int(g.next() % n0)); V = np.concatenate([[2.25], V])
        a.set_batch(I, J, V); b.set_batch(I, J, V)
        mat_equal(a, b, (seed, step, "append batch"))
        for o in (0, 1):
            ia, ib = a.info(o), b.info(o)
            for k in ("stat_extends", "stat_rebalances", "stat_window_slots"):
                assert ia[k] == ib[k], (seed, step, o, k, ia[k], ib[k])
        if g.next() % 2 == 0:                 # delete an older column (never the last: the reference's crash path is a documented divergence)
            jd = 1 + int(g.next() % (col - 1))
            try:
                a.deletecolumn(jd); b.deletecolumn(jd)
            except dsa.DsaError:
                pass
            mat_equal(a, b, (seed, step, "deletecolumn"))
    return "ok"


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    n = 0
    res = {}
    while time.time() - t0 < budget:
        if os.environ.get("FUZZ_ONLY") == "append":      # soak of the append-replay scenario alone
            r = run_append_models(seed)
        elif os.environ.get("FUZZ_ONLY") == "leaf" or (os.environ.get("FUZZ_ONLY") is None and seed % 16 == 7):
            r = run_same_leaf(seed)                         # several count-changing ops per leaf and round
        elif os.environ.get("FUZZ_ONLY") == "leafmat" or (os.environ.get("FUZZ_ONLY") is None and seed % 16 == 15):
            r = run_same_leaf_matrix(seed)                  # ... in a matrix: elements, new columns, emptied columns
        elif os.environ.get("FUZZ_ONLY") == "tomb" or (os.environ.get("FUZZ_ONLY") is None and seed % 16 == 11):
            r = run_tombstones(seed)                        # deletecolumn! / deleterow! and columns next to the tombstones
        elif os.environ.get("FUZZ_ONLY") == "crash" or (os.environ.get("FUZZ_ONLY") is None and seed % 32 == 21):
            r = run_crash_paths(seed)                       # the reference's crash paths: state after a failed batch
        elif os.environ.get("FUZZ_ONLY") == "pcsc" or (os.environ.get("FUZZ_ONLY") is None and seed % 16 == 13):
            r = run_packedcsc(seed)                         # the PackedCSC API with explicit partition ids
        elif os.environ.get("FUZZ_ONLY") == "hot" or (os.environ.get("FUZZ_ONLY") is None and seed % 32 == 1):
            r = run_hot_keys(seed)                          # the same few keys written over and over inside one batch
        elif os.environ.get("FUZZ_ONLY") == "fill" or (os.environ.get("FUZZ_ONLY") is None and seed % 16 == 9):
            r = run_fill(seed)                              # fill mode: addrow!, element appends, closefillmode!, writes behind it
        elif os.environ.get("FUZZ_ONLY") == "shrink" or (os.environ.get("FUZZ_ONLY") is None and seed % 32 == 17):
            r = run_shrink_grow(seed)                       # emptied through several _shrink!s, refilled through _extend!s
        else:
            r = run_shared_words(seed) if seed % 8 == 5 else (run_append_models(seed) if seed % 8 == 3 else (run_matrix(seed) if seed % 4 else run_vector(seed)))
        res[r] = res.get(r, 0) + 1
        n += 1
        seed += 1
        if n % 20 == 0:
            print("fuzz: %d scenarios, %s, %.0f s" % (n, res, time.time() - t0), flush=True)
    print("fuzz done: %d scenarios up to seed %d: %s" % (n, seed - 1, res))
