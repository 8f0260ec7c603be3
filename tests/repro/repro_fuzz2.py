#!/usr/bin/env python3
"""Which ops of a batch touch a region?  Runs the fuzz scenario oracle-vs-oracle (FUZZ_SELF=1, CPU only), records its batches, replays
them on a fresh ORACLE matrix, batch <step> op by op, printing every op that changes the occupancy of the region.
usage: FUZZ_SELF=1 [FUZZ_BIG=1] repro_fuzz2.py <seed> <step> <lo> <hi> (0-based slots, orientation 0)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz
dsa = fuzz.dsa
seed, step, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
batches = []
Mat = dsa.DynamicSparseMatrix
orig_set = Mat.set_batch
first = [None]
def rec(self, I, J, V):
    if first[0] is None: first[0] = self
    if self is first[0]:
        batches.append((np.array(I, dtype=np.int64), np.array(J, dtype=np.int64), np.array(V, dtype=np.float64)))
    return orig_set(self, I, J, V)
Mat.set_batch = rec
print("scenario:", fuzz.run_matrix(seed), len(batches), "batches")
Mat.set_batch = orig_set
b = dsa.dynamicsparse(fill_mode=False, binding=fuzz.ora)
for (I, J, V) in batches[:step]:
    b.set_batch(I, J, V)
I, J, V = batches[step]
np.savez(os.path.join(ROOT, "gpurun_out", "repro_batches_%d.npz" % seed), **{"I%d" % k: x[0] for k, x in enumerate(batches[:step + 1])},
         **{"J%d" % k: x[1] for k, x in enumerate(batches[:step + 1])}, **{"V%d" % k: x[2] for k, x in enumerate(batches[:step + 1])})
print("batch %d:" % step, len(I), "ops; capacity", b.info(0)["capacity"], "segment", b.info(0)["segment_capacity"])
prev = b.export_layout(0)["occ"][lo:hi + 1].copy()
prev_reb = b.info(0)["stat_rebalances"]; prev_ws = b.info(0)["stat_window_slots"]
for t in range(len(I)):
    b.set_batch(I[t:t + 1], J[t:t + 1], V[t:t + 1])
    inf = b.info(0)
    cur = b.export_layout(0)["occ"][lo:hi + 1]
    if not np.array_equal(cur, prev):
        print("op %d: A[%d,%d]=%g  rebalances +%d window slots +%d  region %s -> %s" % (t, I[t], J[t], V[t], inf["stat_rebalances"] - prev_reb, inf["stat_window_slots"] - prev_ws,
              "".join(str(int(x)) for x in prev), "".join(str(int(x)) for x in cur)), flush=True)
        prev = cur.copy()
    prev_reb = inf["stat_rebalances"]; prev_ws = inf["stat_window_slots"]
