#!/usr/bin/env python3
"""What a product costs right after a write batch (dev tool): config 5's loop — 1000 new columns, then y = A x — with the host time of
the SpMV call and of the wait for its result split out."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
dev = torch.device("cuda:0")
m5, ncols5, per5, every = bench.C5_FULL
I5, J5, V5 = bench.c5_columns(m5, ncols5, per5)
x5 = bench.unit12(13, ncols5)
B = dsa.dynamicsparse(fill_mode=False, binding=hip)
xd = torch.from_numpy(x5).to(dev); yd = torch.zeros(m5, dtype=torch.float64, device=dev)
t_call = t_sync = t_second = 0.0; n = 0
for c0 in range(0, ncols5, every):
    sl = slice(c0 * per5, (c0 + every) * per5)
    B.set_batch(I5[sl], J5[sl], V5[sl])
    if "--sync" in sys.argv:      # like bench.py's config-5 leg: the batch's trailing device work (table merge, meta prefetch) is waited for first
        hip.call("mat_sync", B.h)
    t0 = time.perf_counter()
    hip.call("mat_spmv_dense_dev", B.h, 0, 0, C.c_void_p(xd.data_ptr()), c0 + every, C.c_void_p(yd.data_ptr()), m5)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hip.call("mat_spmv_dense_dev", B.h, 0, 0, C.c_void_p(xd.data_ptr()), c0 + every, C.c_void_p(yd.data_ptr()), m5)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    t_call += t1 - t0; t_sync += t2 - t1; t_second += t3 - t2; n += 1
print("product after a write batch: call %.1f us + wait %.1f us = %.1f us ; a second product (meta cached): %.1f us" %
      (t_call / n * 1e6, t_sync / n * 1e6, (t_call + t_sync) / n * 1e6, t_second / n * 1e6))
t0 = time.perf_counter()
for _ in range(200):
    torch.cuda.synchronize()
print("an idle torch.cuda.synchronize(): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
