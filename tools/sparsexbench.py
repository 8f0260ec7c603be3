import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
I,J,V = bench.c3_triplets(1000000, 1000000, 10, 0, 5, 6)
A = dsa.dynamicsparse(I, J, V, 1000000, 1000000, binding=hip)
n3 = 1000000
xi = np.unique(1 + (bench.splitmix_array(50 + 500000, 500000) % np.uint64(n3)).astype(np.int64))
xv = bench.unit12(51, len(xi))
for k in range(8):
    t = time.perf_counter(); yi, yv = A.mul((xi, xv)); dt = time.perf_counter() - t
    print("call %d: %.2f ms  touched %d" % (k, dt*1e3, len(yi)))
