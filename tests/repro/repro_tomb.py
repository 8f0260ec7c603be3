#!/usr/bin/env python3
"""Repro helper for run_tombstones error-code divergences: replays the recorded calls on fresh matrices of both libraries, the failing
batch op by op, and prints the first failing op and its error on each side.  usage: repro_tomb.py <seed>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz
dsa = fuzz.dsa
seed = int(sys.argv[1])
Mat = dsa.DynamicSparseMatrix
calls = []          # (name, args) in call order, recorded on the FIRST matrix object only
first = [None]
def wrap(name):
    orig = getattr(Mat, name)
    def f(self, *a):
        if first[0] is None: first[0] = self
        if self is first[0]: calls.append((name, tuple(np.array(x) if isinstance(x, (list, np.ndarray)) else x for x in a)))
        return orig(self, *a)
    setattr(Mat, name, f)
    return orig
origs = {n: wrap(n) for n in ("set_batch", "deletecolumn", "deleterow")}
build = {}
orig_ds = dsa.dynamicsparse
def ds(*a, **k):
    if "args" not in build: build["args"] = a
    return orig_ds(*a, **k)
fuzz.dsa.dynamicsparse = ds
try:
    print("result:", fuzz.run_tombstones(seed))
except AssertionError as e:
    print("DIVERGED", e)
for n, o in origs.items(): setattr(Mat, n, o)
fuzz.dsa.dynamicsparse = orig_ds
print(len(calls), "calls recorded; last:", calls[-1][0], "with", len(calls[-1][1][0]) if calls[-1][0] == "set_batch" else calls[-1][1])
for lib, name in ((fuzz.hip, "hip"), (fuzz.ora, "oracle")):
    M = orig_ds(*build["args"], binding=lib)
    for nm, args in calls[:-1]:
        try: getattr(M, nm)(*args)
        except dsa.DsaError as e: print(name, "early error", nm, e.code)
    nm, args = calls[-1]
    I, J, V = args
    for t in range(len(I)):
        try:
            M.set_batch(I[t:t + 1], J[t:t + 1], V[t:t + 1])
        except dsa.DsaError as e:
            print("%s: op %d A[%d,%d]=%g fails with code %d: %s" % (name, t, I[t], J[t], V[t], e.code, str(e)[:160]))
            li = M.info(0); lr = M.info(1)
            print("   colmajor table_len %d nb_partitions %d | rowmajor table_len %d nb_partitions %d" % (li["table_len"], li["nb_partitions"], lr["table_len"], lr["nb_partitions"]))
            break
    else:
        print(name, ": no op fails one by one")
