#!/usr/bin/env python3
"""C2 batch B (100k uniform random inserts into a 2^20-slot PMA) and the random matrix updates of bench.py, timed alone (dev tool):
medians of 5 runs, each on a freshly built structure.  A/B: DSA_DEV=1 DSA_RUN_AHEAD=0 / DSA_LIBRARY=<another build>."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
n0 = 700000
keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
vals0 = bench.unit12(3, n0)
odd = np.unique(1 + 2 * (bench.splitmix_array(4, 120000) % np.uint64(700000)).astype(np.int64))[:100000]
np.random.default_rng(4).shuffle(odd)
vb = bench.unit12(4, len(odd))
ts = []
for rep in range(6):
    v = dsa.dynamicsparsevec(keys0, vals0, binding=hip)
    t = time.perf_counter(); v.set_batch(odd, vb); ts.append(time.perf_counter() - t)
    inf = v.info()
    del v
tb = float(np.median(ts[1:]))
print("batch B: %.2f ms  %.1f M inserts/s  rounds %d par ops %d seq ops %d  [%s]" % (tb * 1e3, len(odd) / tb / 1e6, inf["stat_par_rounds"], inf["stat_par_ops"], inf["stat_seq_ops"],
                                                                                  ", ".join("%.2f" % (x * 1e3) for x in ts)))
def pbprof(tag):      # DSA_LIBRARY=<a -DDSA_PB_PROF build of csrc/>: cycle sums per phase of the resolve step (printed to stderr)
    lib = getattr(hip, "lib", None)
    if lib is not None and hasattr(lib, "dsa_dbg_pbprof_dump"):
        sys.stderr.write(tag + " "); sys.stderr.flush(); lib.dsa_dbg_pbprof_dump()
pbprof("batch B")
mm, nn = 20000, 30000
ri = 1 + (bench.splitmix_array(31, 600000) % np.uint64(mm)).astype(np.int64)
ci = 1 + (bench.splitmix_array(32, 600000) % np.uint64(nn)).astype(np.int64)
v0 = bench.unit12(33, 600000)
ui = 1 + (bench.splitmix_array(34, 200000) % np.uint64(mm)).astype(np.int64)
uj = 1 + (bench.splitmix_array(35, 200000) % np.uint64(nn)).astype(np.int64)
uv = np.where(bench.splitmix_array(36, 200000) % np.uint64(4) == 0, 0.0, bench.unit12(37, 200000))
ts = []
for rep in range(6):
    M = dsa.dynamicsparse(ri, ci, v0, mm, nn, binding=hip)
    M.set_batch(ui[:128], uj[:128], uv[:128])
    t = time.perf_counter(); M.set_batch(ui, uj, uv); ts.append(time.perf_counter() - t)
    infs = [M.info(o) for o in (0, 1)]
    del M
tm = float(np.median(ts[1:]))
print("matrix random updates: %.2f ms  %.1f M updates/s  rounds col %d row %d  [%s]" % (tm * 1e3, len(ui) / tm / 1e6, infs[0]["stat_par_rounds"], infs[1]["stat_par_rounds"],
                                                                                     ", ".join("%.2f" % (x * 1e3) for x in ts)))
pbprof("matrix updates")
