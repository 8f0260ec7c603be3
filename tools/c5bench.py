#!/usr/bin/env python3
"""C5 (Coluna-style column streaming) cost split (dev tool).  DSA_DBG_TIME=1 prints the per-orientation time of every batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
full = '--full' in sys.argv
m5, ncols5, per5 = (100_000, 50_000, 16) if full else (10_000, 5_000, 16)
step = int(os.environ.get('C5_STEP', '1000' if full else '500'))     # columns per write batch (Coluna adds a few columns per iteration)
if '--cold' not in sys.argv:      # warm the process (code objects, first graph instantiation, allocator) like bench.py's earlier legs do
    W = dsa.dynamicsparse(fill_mode=False, binding=hip)
    Iw, Jw, Vw = bench.c5_columns(5000, 1250, 16)        # the same kind of stream at 1/40 of the size
    W.set_batch(Iw, Jw, Vw)
    del W
B = dsa.dynamicsparse(fill_mode=False, binding=hip)
rows5 = 1 + (bench.splitmix_array(11, ncols5 * per5 * 2) % np.uint64(m5)).astype(np.int64)
vals5 = bench.unit12(12, ncols5 * per5)
pos = 0; nw = 0; t_w = 0.0; t_d = 0.0; nd = 0
deletes = '--deletes' in sys.argv      # delete 5 % of the streamed columns after every batch (tombstones in the colmajor tables)
nb_max = int(os.environ.get('C5_BATCHES', '1000'))
for c0 in range(0, min(ncols5, nb_max * step), step):
    I5, J5 = [], []
    for j in range(c0 + 1, c0 + step + 1):
        seen = set()
        while len(seen) < per5:
            seen.add(int(rows5[pos])); pos += 1
        rr = sorted(seen)
        I5 += rr; J5 += [j] * per5
    V5 = vals5[nw:nw + len(I5)]
    t = time.perf_counter(); B.set_batch(I5, J5, V5); t_w += time.perf_counter() - t
    nw += len(I5)
    if deletes:
        t = time.perf_counter()
        for j in range(c0 + 1, c0 + step + 1, 20):
            B.deletecolumn(j); nd += 1
        t_d += time.perf_counter() - t
if deletes: print('deletecolumn: %d calls, %.1f ms (%.1f us each)' % (nd, t_d * 1e3, t_d / max(nd, 1) * 1e6))
print("C5 %s:" % ("full" if full else "scaled") + " %d element writes in %.1f ms -> %.0f writes/s" % (nw, t_w * 1e3, nw / t_w))

for o, name in ((0, "colmajor"), (1, "rowmajor")):
    inf = B.info(o)
    print("final info %s:" % name, {k: inf[k] for k in ("capacity", "nb_elements", "stat_rebalances", "stat_window_slots", "stat_extends", "stat_par_rounds", "stat_par_ops", "stat_seq_ops")})
