#!/usr/bin/env python3
"""BASELINE config 2 as bench.py runs it (dev tool): 700k keys in a 2^20-slot PMA, batch A = 100k ascending appends (one extend to 2^21),
batch B = 100k uniform odd keys; medians of 5 runs on freshly built vectors.  A/B: DSA_DEV=1 DSA_LIBRARY=<another build> / DSA_RUN_AHEAD=0."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
n0 = 700000
keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
app = np.arange(1400001, 1500001, dtype=np.int64)
odd = np.unique(1 + 2 * (bench.splitmix_array(4, 120000) % np.uint64(700000)).astype(np.int64))[:100000]
np.random.default_rng(4).shuffle(odd)
vals0, valsA, valsB = bench.unit12(3, n0), bench.unit12(3, 100000), bench.unit12(4, len(odd))
tas, tbs = [], []
for _ in range(6):
    v = dsa.dynamicsparsevec(keys0, vals0, binding=hip)
    t = time.perf_counter(); v.set_batch(app, valsA); tas.append(time.perf_counter() - t)
    r0 = v.info()["stat_par_rounds"]
    t = time.perf_counter(); v.set_batch(odd, valsB); tbs.append(time.perf_counter() - t)
    inf = v.info()
    del v
print("batch A %.2f ms   batch B %.2f ms = %.1f M/s  rounds (B) %d  [%s]" % (np.median(tas[1:]) * 1e3, np.median(tbs[1:]) * 1e3, len(odd) / np.median(tbs[1:]) / 1e6,
                                                                          inf["stat_par_rounds"] - r0, ", ".join("%.2f" % (x * 1e3) for x in tbs)))
