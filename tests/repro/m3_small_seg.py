#!/usr/bin/env python3
"""dev: append runs on vectors GROWN from a few keys (segments of 2 / 8 slots) against the oracle, with timing"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dsa_loader, oracle_binding
dsa = dsa_loader.load(); hip = dsa.product(); ora = oracle_binding.load(dsa)
for n_first, total, runs in [(3, 120000, [40000, 700, 90000, 5000]), (150, 200000, [100000, 100000, 3000]), (1, 70000, [600, 30000])]:
    keys = np.arange(1, total + 1, dtype=np.int64) * 2
    a = dsa.dynamicsparsevec(keys[:n_first], np.ones(n_first), binding=hip)
    b = dsa.dynamicsparsevec(keys[:n_first], np.ones(n_first), binding=ora)
    a.set_batch(keys[n_first:], np.ones(total - n_first)); b.set_batch(keys[n_first:], np.ones(total - n_first))
    nxt = int(keys[-1]) + 1
    for r in runs:
        ks = np.arange(nxt, nxt + r, dtype=np.int64); nxt += r
        t0 = time.perf_counter(); a.set_batch(ks, np.ones(r)); dt = time.perf_counter() - t0
        b.set_batch(ks, np.ones(r))
        la, lb = a.export_layout(), b.export_layout()
        assert np.array_equal(la[2], lb[2]), "bitmap"
        o = la[2].astype(bool)
        assert np.array_equal(la[0][o], lb[0][o]) and np.array_equal(la[1][o], lb[1][o])
        ia, ib = a.info(), b.info()
        for k in ("capacity", "segment_capacity", "stat_extends", "stat_rebalances", "stat_window_slots", "nb_elements"):
            assert ia[k] == ib[k], (k, ia[k], ib[k])
        print("first %d keys, seg %d cap %d: run of %d ok in %.2f ms" % (n_first, ia["segment_capacity"], ia["capacity"], r, dt * 1e3), flush=True)
