import sys, gc
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
def used(): 
    free, tot = torch.cuda.mem_get_info(); return (tot-free)/2**20
I,J,V = bench.c3_triplets(20000, 20000, 10, 0, 5, 6)
base=None
for it in range(120):
    A = dsa.dynamicsparse(I, J, V, 20000, 20000, binding=hip)
    B = dsa.dynamicsparse(fill_mode=True, binding=hip)
    B.set_batch(I[:50000], J[:50000], V[:50000]); B.closefillmode()
    k = 1 + (bench.splitmix_array(it, 3000) % np.uint64(20000)).astype(np.int64)
    A.set_batch(k, k[::-1].copy(), bench.unit12(it, 3000))
    v = dsa.dynamicsparsevec(np.arange(1, 50001, dtype=np.int64)*2, bench.unit12(1, 50000), binding=hip)
    v.set_batch(np.arange(100001, 103001, dtype=np.int64), bench.unit12(2, 3000))
    v.set_batch(k, bench.unit12(3, 3000))
    del A, B, v; gc.collect()
    if it in (19, 119):
        torch.cuda.synchronize(); print(it, "device MiB in use: %.0f" % used())
