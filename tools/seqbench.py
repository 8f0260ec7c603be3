#!/usr/bin/env python3
"""Sequencer cost breakdown (dev tool): find-only, overwrite, insert batches on the C2 vector."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
n0 = 700000
keys0 = np.arange(1, n0 + 1, dtype=np.int64) * 2
v = dsa.dynamicsparsevec(keys0, bench.unit12(3, n0), binding=hip)
rng = np.random.default_rng(1)
def t(label, keys, vals):
    v.set_batch(keys[:100], vals[:100])
    t0 = time.perf_counter(); v.set_batch(keys, vals); dt = time.perf_counter() - t0
    print("%-40s %7.2f us/op" % (label, dt / len(keys) * 1e6))
ex = rng.choice(keys0, 100000)
t("delete of missing keys (find only)", ex + 1 - 2 * (ex % 2 == 1), np.zeros(100000))   # odd keys: absent
t("overwrite existing keys", ex, bench.unit12(9, 100000))
t("1 op batches x200 (host round trip)", ex[:200], bench.unit12(9, 200)) if False else None
t0 = time.perf_counter()
for k in ex[:300]: v[int(k)] = 1.5
print("%-40s %7.2f us/op" % ("single-op calls (host round trip each)", (time.perf_counter() - t0) / 300 * 1e6))
odd = np.unique(1 + 2 * (bench.splitmix_array(4, 120000) % np.uint64(700000)).astype(np.int64))[:100000]
rng.shuffle(odd)
t("uniform inserts (new keys)", odd, bench.unit12(4, len(odd)))
print({k: v.info()[k] for k in ("stat_par_rounds", "stat_par_ops", "stat_seq_ops")})
t("delete them again", odd, np.zeros(len(odd)))

# matrix: random updates on an existing 20k x 30k structure (both orientations), batch-parallel vs sequential
import subprocess
m, n = 20000, 30000
rows = 1 + (bench.splitmix_array(31, 600000) % np.uint64(m)).astype(np.int64)
cols = 1 + (bench.splitmix_array(32, 600000) % np.uint64(n)).astype(np.int64)
A = dsa.dynamicsparse(rows, cols, bench.unit12(33, 600000), m, n, binding=hip)
ui = 1 + (bench.splitmix_array(34, 200000) % np.uint64(m)).astype(np.int64)
uj = 1 + (bench.splitmix_array(35, 200000) % np.uint64(n)).astype(np.int64)
uv = np.where(bench.splitmix_array(36, 200000) % np.uint64(4) == 0, 0.0, bench.unit12(37, 200000))
A.set_batch(ui[:100], uj[:100], uv[:100])
t0 = time.perf_counter(); A.set_batch(ui, uj, uv); dt = time.perf_counter() - t0
print("%-40s %7.2f us/op (%s)" % ("matrix random A[i,j]=v (2 PCSR writes)", dt / len(ui) * 1e6, os.environ.get("DSA_PARBATCH", "1")))
# small batches on the same matrix: latency of one set_batch call by batch size
for nb in (16, 130, 500, 2000, 8000):
    reps = 20
    off = 1000
    t0 = time.perf_counter()
    for r in range(reps):
        sl = slice(off + r * nb, off + (r + 1) * nb)
        A.set_batch(ui[sl], uj[sl], uv[sl])
    dt = (time.perf_counter() - t0) / reps
    print("matrix set_batch of %5d writes: %8.1f us per call  %6.2f us/op" % (nb, dt * 1e6, dt / nb * 1e6))
