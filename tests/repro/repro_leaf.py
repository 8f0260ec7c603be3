#!/usr/bin/env python3
"""Repro helper for run_same_leaf: prints where the layouts diverge.  usage: repro_leaf.py <seed>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz
dsa = fuzz.dsa
seed = int(sys.argv[1])
Vec = dsa.DynamicSparseVector
orig = Vec.set_batch
state = {"objs": [], "batches": []}
def rec(self, k, v):
    if not any(self is o for o in state["objs"]): state["objs"].append(self)
    if self is state["objs"][0]: state["batches"].append((np.array(k), np.array(v)))
    return orig(self, k, v)
Vec.set_batch = rec
try:
    print("result:", fuzz.run_same_leaf(seed))
except AssertionError as e:
    print("DIVERGED", e)
    a, b = state["objs"][0], state["objs"][1]
    ka, kb = a.export_layout(), b.export_layout()
    ia, ib = a.info(), b.info()
    print({k: (ia[k], ib.get(k)) for k in ("capacity", "segment_capacity", "nb_elements", "stat_rebalances", "stat_window_slots", "stat_par_rounds", "stat_par_ops", "stat_seq_ops")})
    d = np.nonzero(ka[2] != kb[2])[0]
    print("occ differs at %d slots:" % len(d), d[:24])
    if len(d):
        lo, hi = int(d[0]) - 24, int(d[-1]) + 24
        lo = max(lo, 0)
        print("region [%d, %d) 0-based; hip then oracle:" % (lo, hi))
        print("".join(str(int(x)) for x in ka[2][lo:hi]))
        print("".join(str(int(x)) for x in kb[2][lo:hi]))
        keys, vals = state["batches"][-1]
        Kb = kb[0]
        klo, khi = int(Kb[lo:hi][kb[2][lo:hi].astype(bool)].min()), int(Kb[lo:hi][kb[2][lo:hi].astype(bool)].max())
        near = [(i, int(k), float(v)) for i, (k, v) in enumerate(zip(keys, vals)) if klo - 40 <= k <= khi + 40]
        print("ops of the last batch (%d) with keys near the region [%d, %d]:" % (len(keys), klo, khi), near)
        np.savez(os.path.join(ROOT, "gpurun_out", "repro_leaf_%d.npz" % seed), lo=lo, hi=hi, **{"k%d" % i: x[0] for i, x in enumerate(state["batches"])}, **{"v%d" % i: x[1] for i, x in enumerate(state["batches"])})
