#!/usr/bin/env python3
"""Fill-mode flush timing (dev tool): the C3 triples through the fill buffer in 10 batches, then closefillmode!; twice (the
second matrix reuses the cached pinned staging chunks)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
I, J, V = bench.c3_triplets(1000000, 1000000, 10, 0, 5, 6)
for rep in range(3):
    A = dsa.dynamicsparse(fill_mode=True, binding=hip)
    t = time.perf_counter()
    for c in range(0, len(I), 1000000):
        A.set_batch(I[c:c + 1000000], J[c:c + 1000000], V[c:c + 1000000])
    t1 = time.perf_counter()
    A.closefillmode()
    t2 = time.perf_counter()
    print("fill mode: 10 batches of 1M triples %.1f ms, closefillmode (last chunk + build) %.2f ms, nnz %d" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, A.nnz()))
    del A
t = time.perf_counter()
B = dsa.dynamicsparse(I, J, V, 1000000, 1000000, binding=hip)
print("dynamicsparse(I, J, V) from caller memory: %.1f ms" % ((time.perf_counter() - t) * 1e3))
