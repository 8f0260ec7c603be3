import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
k50 = np.arange(1, 51, dtype=np.int64) * 7; v50 = bench.unit12(3, 50)
for _ in range(5): dsa.dynamicsparsevec(k50, v50, binding=hip)
os.environ["X"]="1"
t=time.perf_counter()
v = dsa.dynamicsparsevec(k50, v50, binding=hip)
print("create %.1f us" % ((time.perf_counter()-t)*1e6))
t=time.perf_counter(); del v; print("destroy %.1f us" % ((time.perf_counter()-t)*1e6))
