#!/usr/bin/env python3
"""Kernel micro-bench on the C3 matrix (dev tool): times the SpMV and the root rebalance with HIP
events on the launch stream.  Usage: python tools/kbench.py [--small]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dsa_loader  # noqa: E402

dsa = dsa_loader.load()
hip = dsa.product()
small = "--small" in sys.argv
m = ncl = 250_000 if small else 1_000_000
I, J, V = bench.c3_triplets(m, ncl, 10, 0, 5, 6)
t = time.time()
A = dsa.dynamicsparse(I, J, V, m, ncl, binding=hip)
print("build %.2fs cap %d" % (time.time() - t, A.info(1)["capacity"]))
stream = torch.cuda.current_stream()
hip.call("mat_set_stream", A.h, C.c_void_p(stream.cuda_stream))
dev = torch.device("cuda:0")
x = torch.from_numpy(bench.unit12(7, ncl)).to(dev)
y = torch.zeros(m, dtype=torch.float64, device=dev)
cap = A.info(1)["capacity"]


def timeit(fn, reps=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


for label, algo, tr, nx in (("gather A*x", 0, 0, ncl), ("gather A*x, nx=0 (no x gather)", 0, 0, 0),
                            ("gather A'*x", 0, 1, m), ("scatter A*x", 1, 0, ncl)):
    us = timeit(lambda: hip.call("mat_spmv_dense_dev", A.h, tr, algo, C.c_void_p(x.data_ptr()), nx, C.c_void_p(y.data_ptr()), m))
    b = 16 * cap + 8 * ncl + 8 * m
    print("%-34s %8.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)" % (label, us, b / us / 1e3, b / us / 1e3 / 80))
for o in (0, 1):
    us = timeit(lambda: A.rebalance_root(o), reps=20)
    b = 32 * cap
    print("rebalance root orient %d            %8.1f us  %7.1f GB/s (%.1f%%)" % (o, us, b / us / 1e3, b / us / 1e3 / 80))
ys = torch.zeros(m, dtype=torch.float64, device=dev)
us = timeit(lambda: ys.copy_(y))
print("torch copy 8MB %.1f us" % us)
big = torch.empty(cap * 2, dtype=torch.float64, device=dev)
big2 = torch.empty(cap * 2, dtype=torch.float64, device=dev)
us = timeit(lambda: big2.copy_(big))
print("torch D2D copy %d MB: %.1f us -> %.1f GB/s (read+write)" % (cap * 16 >> 20, us, 2 * cap * 16 / us / 1e3))

us = timeit(lambda: big2.zero_())
print("torch memset %d MB: %.1f us -> %.1f GB/s (write only)" % (cap * 16 >> 20, us, cap * 16 / us / 1e3))
us = timeit(lambda: big.sum())
print("torch sum (read only) %d MB: %.1f us -> %.1f GB/s" % (cap * 16 >> 20, us, cap * 16 / us / 1e3))

# random 8-byte gather rate of the hardware for different table sizes (torch index_select)
for nx_, label in ((1_000_000, "8 MB"), (500_000, "4 MB"), (250_000, "2 MB"), (125_000, "1 MB"), (4_000_000, "32 MB")):
    xt = torch.rand(nx_, dtype=torch.float64, device=dev)
    idx = torch.randint(0, nx_, (11_000_000,), device=dev, dtype=torch.int64)
    us = timeit(lambda: torch.index_select(xt, 0, idx))
    print("torch gather 11M x f64 from %s table: %.1f us -> %.1f G gathers/s" % (label, us, 11e6 / us / 1e3))

# banded matrix (|row - col| < 4096): the XCD-aware tile map keeps each XCD's x range L2-resident
import numpy as np
nb = 1_000_000
colsb = np.repeat(np.arange(1, nb + 1, dtype=np.int64), 10)
offs = (bench.splitmix_array(21, nb * 10) % np.uint64(8192)).astype(np.int64) - 4096
rowsb = np.clip(colsb + offs, 1, nb)
keyb = colsb * np.int64(nb + 1) + rowsb
_, fb = np.unique(keyb, return_index=True)
Bm = dsa.dynamicsparse(rowsb[fb], colsb[fb], bench.unit12(22, len(fb)), nb, nb, binding=hip)
hip.call("mat_set_stream", Bm.h, C.c_void_p(stream.cuda_stream))
capb = Bm.info(1)["capacity"]
us = timeit(lambda: hip.call("mat_spmv_dense_dev", Bm.h, 0, 0, C.c_void_p(x.data_ptr()), nb, C.c_void_p(y.data_ptr()), nb))
bb = 16 * capb + 16 * nb
print("banded 1M x 1M, ~10M nnz: gather A*x   %8.1f us  %7.1f GB/s (%.1f%%)" % (us, bb / us / 1e3, bb / us / 1e3 / 80))
