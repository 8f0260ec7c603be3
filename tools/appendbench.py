#!/usr/bin/env python3
"""Append-run cost breakdown (dev tool): BASELINE config 2 batch A on the C2 vector.  DSA_DBG_RUN=1 prints the in-kernel profile."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
n0 = 700000
keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
for rep in range(2):
    v = dsa.dynamicsparsevec(keys0, bench.unit12(3, n0), binding=hip)
    app = np.arange(1400001, 1500001, dtype=np.int64)
    va = bench.unit12(3, 100000)
    t0 = time.perf_counter(); v.set_batch(app, va); dt = time.perf_counter() - t0
    print("batch A: %.2f ms  %.0f appends/s" % (dt * 1e3, len(app) / dt), v.info()["capacity"])

if "--check" in sys.argv:      # parity of the batch against the CPU oracle (used by the test-suite under DSA_POS_WIDE=1)
    sys.path.insert(0, os.path.join(ROOT, "oracle")); import oracle_binding; ora = oracle_binding.load(dsa)
    o = dsa.dynamicsparsevec(keys0, bench.unit12(3, n0), binding=ora)
    o.set_batch(app, va)
    a, b = v.export_layout(), o.export_layout()
    assert np.array_equal(a[2], b[2])
    occ = a[2].astype(bool)
    assert np.array_equal(a[0][occ], b[0][occ]) and np.array_equal(a[1][occ], b[1][occ])
    assert v.info()["stat_window_slots"] == o.info()["stat_window_slots"], (v.info(), o.info())
    print("append parity ok")
