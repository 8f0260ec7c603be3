#!/usr/bin/env python3
"""One orientation's K-build alone (dev tool): dynamicsparsevec of 10 M random keys — the kernels of csrc/build.hip on ONE stream, for
per-kernel times that are not blurred by the twin orientation's build (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
n = 10_000_000
keys = 1 + (bench.splitmix_array(71, n) % np.uint64(1 << 40)).astype(np.int64)
vals = bench.unit12(72, n)
for rep in range(4):
    t = time.perf_counter()
    v = dsa.dynamicsparsevec(keys, vals, binding=hip)
    dt = time.perf_counter() - t
    print("dynamicsparsevec of %d keys: %.1f ms (nnz %d)" % (n, dt * 1e3, v.nnz()))
    del v
