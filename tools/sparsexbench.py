#!/usr/bin/env python3
"""Sparse-x product through the host-pointer entry point on the C3 matrix (dev tool): 100 / 10 k / 500 k stored entries, per-call times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
I, J, V = bench.c3_triplets(1000000, 1000000, 10, 0, 5, 6)
A = dsa.dynamicsparse(I, J, V, 1000000, 1000000, binding=hip)
n3 = 1000000
if "--torch-stream" in sys.argv:      # like bench.py: the matrix bound to torch's current stream
    import ctypes as C
    import torch
    hip.call("mat_set_stream", A.h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    print("bound to torch stream", torch.cuda.current_stream().cuda_stream)
if "--rebalance-first" in sys.argv:   # like bench.py: root rebalances of the colmajor orientation before the products
    for _ in range(5):
        A.rebalance_root(dsa.COLMAJOR)
for nxs in (100, 10_000, 500_000):
    xi = np.unique(1 + (bench.splitmix_array(50 + nxs, nxs) % np.uint64(n3)).astype(np.int64))
    xv = bench.unit12(51, len(xi))
    ts = []
    for k in range(6):
        t = time.perf_counter(); yi, yv = A.mul((xi, xv)); ts.append((time.perf_counter() - t) * 1e3)
    print("stored %d touched %d: calls [%s] ms" % (len(xi), len(yi), ", ".join("%.3f" % x for x in ts)))
    bufs = (np.zeros(n3, dtype=np.int64), np.zeros(n3))          # result arrays the caller keeps (pages already faulted in)
    ts = []
    for k in range(6):
        t = time.perf_counter(); yi, yv = A.mul((xi, xv), out=bufs); ts.append((time.perf_counter() - t) * 1e3)
    print("   into result arrays the caller keeps: calls [%s] ms" % ", ".join("%.3f" % x for x in ts))
    try:                      # every operand in HBM: HIP events around 10 enqueued products (no host wait inside)
        import torch
        dev = torch.device("cuda")
        d_xi = torch.from_numpy(xi).to(dev); d_xv = torch.from_numpy(xv).to(dev)
        d_yi = torch.empty(n3, dtype=torch.int64, device=dev); d_yv = torch.empty(n3, dtype=torch.float64, device=dev); d_c = torch.zeros(1, dtype=torch.int64, device=dev)
        import ctypes as C
        hip.call("mat_set_stream", A.h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        f = lambda: A.mul_dev(d_xi.data_ptr(), d_xv.data_ptr(), len(xi), d_yi.data_ptr(), d_yv.data_ptr(), n3, d_c.data_ptr())
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print("   dsa_mat_spmv_sparse_dev: %.1f us per product (count %d)" % (e0.elapsed_time(e1) * 100, int(d_c.item())))
    except ImportError:
        pass
