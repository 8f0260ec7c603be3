#!/usr/bin/env python3
"""Launch sequence for the PMC passes (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
  3 x SpMV gather (C3)            -> k_spmv dispatches 0..2
  3 x SpMV gather with nx = 0     -> k_spmv dispatches 3..5  (streams only: calibrates the 8 B/lane coalesced pattern)
  3 x root rebalance (colmajor)   -> k_move2<false> dispatches (the LAST 3 of the run)
The same kernels as bench.py on the same C3 matrix; no timing here."""
import ctypes as C
import os
import sys

os.environ["DSA_DEV"] = "1"
os.environ["DSA_SPMV_STREAM"] = "nt"      # the nx = 0 calibration launches must be the SAME kernel instantiation (non-temporal stream loads)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dsa_loader  # noqa: E402

dsa = dsa_loader.load()
hip = dsa.product()
m = ncl = 1_000_000
I, J, V = bench.c3_triplets(m, ncl, 10, 0, 5, 6)
A = dsa.dynamicsparse(I, J, V, m, ncl, binding=hip)
dev = torch.device("cuda:0")
hip.call("mat_set_stream", A.h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
x = torch.from_numpy(bench.unit12(7, ncl)).to(dev)
y = torch.zeros(m, dtype=torch.float64, device=dev)
for nx in (ncl, ncl, ncl, 0, 0, 0):
    hip.call("mat_spmv_dense_dev", A.h, 0, 0, C.c_void_p(x.data_ptr()), nx, C.c_void_p(y.data_ptr()), m)
    torch.cuda.synchronize()
for _ in range(3):
    A.rebalance_root(dsa.COLMAJOR)
    torch.cuda.synchronize()
print("cap", A.info(1)["capacity"])
