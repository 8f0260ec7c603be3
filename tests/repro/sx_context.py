#!/usr/bin/env python3
"""Why is the 100-entry sparse-x product 0.75 ms inside bench.py and 0.09 ms alone?  Runs bench.py's legs one at a time in front of it."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
dev = torch.device("cuda:0")
I, J, V = bench.c3_triplets(1000000, 1000000, 10, 0, 5, 6)
A = dsa.dynamicsparse(I, J, V, 1000000, 1000000, binding=hip)
hip.call("mat_set_stream", A.h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
xi = np.unique(1 + (bench.splitmix_array(150, 100) % np.uint64(1000000)).astype(np.int64)); xv = bench.unit12(51, len(xi))
def probe(tag):
    ts = []
    for k in range(6):
        t = time.perf_counter(); A.mul((xi, xv)); ts.append((time.perf_counter() - t) * 1e3)
    print("%-28s [%s] ms" % (tag, ", ".join("%.3f" % x for x in ts)), flush=True)
probe("fresh")
x = torch.from_numpy(bench.unit12(7, 1000000)).to(dev); y = torch.zeros(1000000, dtype=torch.float64, device=dev)
for _ in range(30):
    hip.call("mat_spmv_dense_dev", A.h, 0, 0, C.c_void_p(x.data_ptr()), 1000000, C.c_void_p(y.data_ptr()), 1000000)
torch.cuda.synchronize(); probe("after dense products")
for _ in range(10): A.rebalance_root(dsa.COLMAJOR)
torch.cuda.synchronize(); probe("after root rebalances")
v = dsa.dynamicsparsevec(np.arange(1, 700001, dtype=np.int64) * 3, bench.unit12(40, 700000), binding=hip)
hip.call("vec_set_stream", v.h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
for _ in range(10): v.rebalance_root()
hip.call("vec_dev_relayout", v.h, 1); v.rebalance_root(); torch.cuda.synchronize(); del v
probe("after vector sweeps")
m4, n4 = 10_000_000, 1_250_000
I4, J4, V4 = bench.c3_triplets(m4, n4, 10, 0, 8, 9)
A4 = dsa.dynamicsparse(I4, J4, V4, m4, n4, binding=hip)
probe("after building the C4 shard")
hip.call("mat_set_stream", A4.h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
x4 = torch.from_numpy(bench.unit12(10, n4)).to(dev); y4 = torch.zeros(m4, dtype=torch.float64, device=dev)
for _ in range(5):
    hip.call("mat_spmv_dense_dev", A4.h, 0, 0, C.c_void_p(x4.data_ptr()), n4, C.c_void_p(y4.data_ptr()), m4)
torch.cuda.synchronize(); probe("after C4 products")
del A4, x4, y4
probe("after freeing the C4 shard")
# the banded leg of bench.py, then the probe again; then the pieces of A.mul one by one
mb = nb = 1_000_000
z = bench.splitmix_array(51, nb * 10)
colb = np.repeat(np.arange(1, nb + 1, dtype=np.int64), 10)
rowb = np.clip(colb + (z % np.uint64(8192)).astype(np.int64) - 4096, 1, mb)
keyb = colb * np.int64(mb + 1) + rowb
_, firstb = np.unique(keyb, return_index=True)
Ab = dsa.dynamicsparse(rowb[firstb], colb[firstb], bench.unit12(52, len(firstb)), mb, nb, binding=hip)
probe("after building the banded matrix")
del Ab
probe("after freeing it")
import gc
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold(), "objects", len(gc.get_objects()))
gc.disable(); probe("gc disabled"); gc.enable()
