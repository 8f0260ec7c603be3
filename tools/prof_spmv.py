#!/usr/bin/env python3
"""Launch sequence for counter passes on the SpMV alone (run under rocprofv3 --pmc ...): 3 x gather SpMV on the C3 matrix,
then 3 x the same with nx = 0 (no x gather).  No timing here."""
import ctypes as C
import os
import sys

os.environ["DSA_DEV"] = "1"
os.environ["DSA_SPMV_STREAM"] = "nt"      # the nx = 0 launches must be the SAME kernel instantiation as the real ones

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
if "cold" in sys.argv[1:]:
    # the gather kernel COLD (kernel-trace passes): every launch behind a 1 GiB device write, as bench.py's roofline.cold_* numbers
    scr = torch.empty((1 << 30) // 4, dtype=torch.float32, device=dev)
    for i in range(12):
        scr.fill_(float(i))
        hip.call("mat_spmv_dense_dev", A.h, 0, 0, C.c_void_p(x.data_ptr()), ncl, C.c_void_p(y.data_ptr()), m)
        torch.cuda.synchronize()
    print("cap", A.info(1)["capacity"], "cold launches 12")
    sys.exit(0)
for nx in (ncl, ncl, ncl, 0, 0, 0):
    hip.call("mat_spmv_dense_dev", A.h, 0, 0, C.c_void_p(x.data_ptr()), nx, C.c_void_p(y.data_ptr()), m)
    torch.cuda.synchronize()
# k_spmv_meta alone at 1 M table entries: a one-element write batch re-arms the prefetch on both orientations (the two launches
# above ran beside the other orientation's bulk build)
for _ in range(3):
    A.set_batch([1], [1], [2.5])
    hip.call("mat_sync", A.h)
print("cap", A.info(1)["capacity"])
