#!/usr/bin/env python3
"""Dev / test tool: dense products of a set of seeded matrices, written to an .npz — run once per DSA_SPMV_SHARE setting (the knob is read
when the library first launches the kernel) and compare the files bit for bit (tests/test_hip_parity.py).
usage: spmv_sharecheck.py <out.npz>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()


def cases():
    # (name, I, J, V, m, n): row lengths from 1 to beyond a 512-slot span, runs of empty rows, tiny and tile-sized capacities
    out = []
    for seed, (m, n, per) in enumerate([(40, 30, 2), (700, 900, 3), (5000, 4000, 9), (60000, 50000, 10), (300, 200000, 1)]):
        I = 1 + (bench.splitmix_array(100 + seed, n * per) % np.uint64(m)).astype(np.int64)
        J = np.repeat(np.arange(1, n + 1, dtype=np.int64), per)
        out.append(("uniform%d" % seed, I, J, bench.unit12(200 + seed, n * per), m, n))
    # ragged: row r holds (r * 37) % 701 cells -> rows that end in the word behind a span, on a span boundary, across tiles
    I, J = [], []
    for r in range(1, 1500):
        k = (r * 37) % 701
        I += [r] * k; J += list(range(1, k + 1))
    out.append(("ragged", np.array(I, dtype=np.int64), np.array(J, dtype=np.int64), bench.unit12(300, len(I)), 1500, 701))
    # very long rows next to single cells
    n = 20000
    I = np.concatenate([np.full(n, 1), np.arange(2, 2002), np.full(n // 3, 5000)]).astype(np.int64)
    J = np.concatenate([np.arange(1, n + 1), np.arange(1, 2001), np.arange(1, n + 1, 3)[: n // 3]]).astype(np.int64)
    out.append(("long", I, J, bench.unit12(301, len(I)), 5000, n))
    return out


res = {}
for name, I, J, V, m, n in cases():
    A = dsa.dynamicsparse(I, J, V, m, n, binding=hip)
    res[name + "_y"] = A.mul(bench.unit12(400, n))
    res[name + "_yt"] = A.mul(bench.unit12(401, m), transpose=True)
    res[name + "_cap"] = np.array([A.info(0)["capacity"], A.info(1)["capacity"]])
    # streamed writes behind the build move cells and leave tombstone-free tables: product again
    A.set_batch(I[: len(I) // 7], J[: len(I) // 7], np.zeros(len(I) // 7))
    res[name + "_y2"] = A.mul(bench.unit12(400, n))
    del A
np.savez(sys.argv[1], **res)
print("sharecheck wrote %d arrays" % len(res))
