#!/usr/bin/env python3
"""Latency of m[:, j] / m[i, :] (new device vectors built device to device) and of the device-resident views (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
rng = np.random.default_rng(5)
m, n = 200000, 1000
sizes = {1: 16, 2: 1000, 3: 10000, 4: 16000, 5: 100000}
I = np.concatenate([rng.choice(m, c, replace=False) + 1 for c in sizes.values()] + [rng.integers(1, m + 1, 50000)])
J = np.concatenate([np.full(c, j) for j, c in sizes.items()] + [rng.integers(6, n + 1, 50000)])
A = dsa.dynamicsparse(I, J, rng.random(len(I)) + 1.0, m, n, binding=hip)
def lat(fn, reps=300):
    for _ in range(10): fn()
    t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e6
for j, c in sizes.items():
    keep = []
    for _ in range(10): A.col_slice(j)
    t = time.perf_counter()
    for _ in range(300): keep.append(A.col_slice(j))          # the handles stay alive: creation alone
    t1 = time.perf_counter()
    del keep                                                   # teardown: a stream wait for the (asynchronous) spread + frees per vector
    t2 = time.perf_counter()
    print("col_slice of %6d entries: %.1f us per call, teardown %.1f us" % (c, (t1 - t) / 300 * 1e6, (t2 - t1) / 300 * 1e6))
try:
    import torch
    dk = torch.empty(200000, dtype=torch.int64, device="cuda"); dv = torch.empty(200000, dtype=torch.float64, device="cuda")
    for j, c in sizes.items():
        print("col_view_dev of %6d entries: %.1f us" % (c, lat(lambda: A.col_view_dev(j, dk.data_ptr(), dv.data_ptr(), 200000))))
except ImportError:
    pass
print("row_slice: %.1f us" % lat(lambda: A.row_slice(int(I[0]))))
