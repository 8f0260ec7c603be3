import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
keys = np.unique(1 + (bench.splitmix_array(1, 12000) % np.uint64(10**7)).astype(np.int64))[:10000]
for rep in range(3):
    v = dsa.dynamicsparsevec(keys, bench.unit12(2, len(keys)), binding=hip)
    newk = 1 + (bench.splitmix_array(7 + rep, 1000) % np.uint64(10**7)).astype(np.int64)
    t = time.perf_counter(); v.set_batch(newk, bench.unit12(9, 1000)); dt = time.perf_counter() - t
    inf = v.info()
    print("C1-like: 1000 random writes on a 10k vector: %.2f ms (%.0f ops/s) rounds %d par %d seq %d" % (dt*1e3, 1000/dt, inf["stat_par_rounds"], inf["stat_par_ops"], inf["stat_seq_ops"]))
