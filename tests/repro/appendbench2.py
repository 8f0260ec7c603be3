#!/usr/bin/env python3
"""dev: batch A three times — fresh vector while the old one is alive (new stream), then after deleting the old one (pooled stream)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
n0 = 700000
keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
v = None
for rep in range(6):
    if rep >= 3 and v is not None:
        v.close(); v = None
    w = dsa.dynamicsparsevec(keys0, bench.unit12(3, n0), binding=hip)
    app = np.arange(1400001, 1500001, dtype=np.int64)
    va = bench.unit12(3, 100000)
    t0 = time.perf_counter(); w.set_batch(app, va); dt = time.perf_counter() - t0
    print("rep %d (%s): batch A %.2f ms  %.0f appends/s" % (rep, "old vector closed first" if rep >= 3 else "old vector alive", dt * 1e3, len(app) / dt), flush=True)
    v = w
