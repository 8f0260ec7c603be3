#!/usr/bin/env python3
"""Latency of scalar lookups and column views through the host-pointer ABI (dev tool): v[k], A[i, j], view(A, :, j)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
n0 = 700000
v = dsa.dynamicsparsevec(np.arange(1, n0 + 1, dtype=np.int64) * 2, bench.unit12(3, n0), binding=hip)
I, J, V = bench.c3_triplets(100000, 100000, 10, 0, 5, 6)
A = dsa.dynamicsparse(I, J, V, 100000, 100000, binding=hip)
def lat(fn, reps=2000):
    for _ in range(20): fn()
    t = time.perf_counter()
    for k in range(reps): fn(k)
    return (time.perf_counter() - t) / reps * 1e6
print("v[k]          %.1f us" % lat(lambda k=0: v[2 * (k % n0) + 2]))
print("A[i, j]       %.1f us" % lat(lambda k=0: A[int(I[k]), int(J[k])]))
print("col view      %.1f us" % lat(lambda k=0: A.col_view(1 + k % 100000)))
print("get_batch(16) %.1f us" % lat(lambda k=0: A.get_batch(I[k:k + 16], J[k:k + 16])))
def setget(k=0):
    A[int(I[k]), int(J[k])] = 2.5          # an overwrite (write-combined), flushed by the read behind it
    return A[int(I[k]), int(J[k])]
print("A[i,j]=v; A[i,j]   %.1f us" % lat(setget, 1000))
def vsetget(k=0):
    v[2 * (k % n0) + 2] = 3.5
    return v[2 * (k % n0) + 2]
print("v[k]=x; v[k]       %.1f us" % lat(vsetget, 1000))
def newelem(k=0):
    A[1 + (k * 7919) % 100000, 1 + (k * 104729) % 100000] = 1.5     # mostly new elements of existing columns
    return A[1 + (k * 7919) % 100000, 1 + (k * 104729) % 100000]
print("new A[i,j]=v; read %.1f us" % lat(newelem, 1000))
