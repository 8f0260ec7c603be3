#!/usr/bin/env python3
"""C2 batch B (100k uniform random inserts into a 2^20-slot PMA) and the random matrix updates of bench.py, timed alone (dev tool)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
for rep in range(2):
    n0 = 700000
    keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
    v = dsa.dynamicsparsevec(keys0, bench.unit12(3, n0), binding=hip)
    odd = np.unique(1 + 2 * (bench.splitmix_array(4, 120000) % np.uint64(700000)).astype(np.int64))[:100000]
    np.random.default_rng(4).shuffle(odd)
    t = time.perf_counter(); v.set_batch(odd, bench.unit12(4, len(odd))); tb = time.perf_counter() - t
    inf = v.info()
    print("batch B: %.2f ms  %.0f inserts/s  rounds %d par ops %d seq ops %d" % (tb * 1e3, len(odd) / tb, inf["stat_par_rounds"], inf["stat_par_ops"], inf["stat_seq_ops"]))
mm, nn = 20000, 30000
ri = 1 + (bench.splitmix_array(31, 600000) % np.uint64(mm)).astype(np.int64)
ci = 1 + (bench.splitmix_array(32, 600000) % np.uint64(nn)).astype(np.int64)
M = dsa.dynamicsparse(ri, ci, bench.unit12(33, 600000), mm, nn, binding=hip)
ui = 1 + (bench.splitmix_array(34, 200000) % np.uint64(mm)).astype(np.int64)
uj = 1 + (bench.splitmix_array(35, 200000) % np.uint64(nn)).astype(np.int64)
uv = np.where(bench.splitmix_array(36, 200000) % np.uint64(4) == 0, 0.0, bench.unit12(37, 200000))
M.set_batch(ui[:128], uj[:128], uv[:128])
t = time.perf_counter(); M.set_batch(ui, uj, uv); tm = time.perf_counter() - t
print("matrix random updates: %.2f ms  %.0f updates/s" % (tm * 1e3, len(ui) / tm))
