import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT","."))
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
n0 = 700000
keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
v = dsa.dynamicsparsevec(keys0, bench.unit12(3, n0), binding=hip)
odd = np.unique(1 + 2 * (bench.splitmix_array(4, 120000) % np.uint64(700000)).astype(np.int64))[:100000]
np.random.default_rng(4).shuffle(odd)
v.set_batch(odd, bench.unit12(4, len(odd)))
