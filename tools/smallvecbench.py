#!/usr/bin/env python3
"""Many small vectors (dev tool): what creating, writing to, reading and destroying a small DynamicSparseVector costs through the ABI."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
def us(fn, reps):
    fn(); fn()
    t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e6
k50 = np.arange(1, 51, dtype=np.int64) * 7; v50 = bench.unit12(3, 50)
print("create + destroy a 50-entry vector      %.1f us" % us(lambda: dsa.dynamicsparsevec(k50, v50, binding=hip), 300))
print("create + destroy an empty vector        %.1f us" % us(lambda: dsa.dynamicsparsevec([], [], binding=hip), 300))
vs = [dsa.dynamicsparsevec(k50, v50, binding=hip) for _ in range(200)]
it = iter(range(10**9))
def write_one():
    i = next(it); vs[i % 200][1000 + i] = 2.5; return vs[i % 200][1000 + i]
print("v[k] = x; v[k] on 200 vectors in turn   %.1f us" % us(write_one, 600))
def batch16():
    i = next(it); vs[i % 200].set_batch(np.arange(5000 + 16 * i, 5016 + 16 * i, dtype=np.int64), v50[:16])
print("set_batch of 16 new keys                %.1f us" % us(batch16, 400))
rng = np.random.default_rng(5)
def batch16r():
    i = next(it); vs[i % 200].set_batch(rng.integers(1, 10**6, 16), v50[:16])
print("set_batch of 16 scattered keys          %.1f us" % us(batch16r, 400))
print("nonzeros() of a ~100-entry vector       %.1f us" % us(lambda: vs[next(it) % 200].nonzeros(), 400))
a, b = vs[0], vs[1]
print("v1 + v2                                 %.1f us" % us(lambda: a + b, 200))
print("v1 == v2                                %.1f us" % us(lambda: a == b, 200))
