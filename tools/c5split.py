"""dev: per batch of config 5 at full size, when each orientation finished (DSA_DEV=1 DSA_DBG_TIME=1 prints them): what a run costs as the sum over
the batches of max(colmajor, rowmajor) — the library joins both per batch — against max(sum colmajor, sum rowmajor), the bound of two decoupled pipelines."""
import os, re, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
m5, ncols5, per5, every = bench.C5_FULL
I5, J5, V5 = bench.c5_columns(m5, ncols5, per5)
B = dsa.dynamicsparse(fill_mode=False, binding=hip)
for c0 in range(0, ncols5, every):
    sl = slice(c0 * per5, (c0 + every) * per5)
    B.set_batch(I5[sl], J5[sl], V5[sl]); hip.call("mat_sync", B.h)
''' % ROOT
env = dict(os.environ, DSA_DEV="1", DSA_DBG_TIME="1")
out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stderr
rows = re.findall(r"both orientations ([0-9.]+) ms \(colmajor done after ([0-9.]+) ms, rowmajor after ([0-9.]+) ms\)", out)
tot = [float(a) for a, _, _ in rows]; col = [float(b) for _, b, _ in rows]; row = [float(c) for _, _, c in rows]
print("batches %d | joined per batch: %.1f ms | colmajor alone %.1f ms, rowmajor alone %.1f ms -> decoupled bound %.1f ms" % (len(rows), sum(tot), sum(col), sum(row), max(sum(col), sum(row))))
print("colmajor per batch:", " ".join("%.1f" % x for x in col))
print("rowmajor per batch:", " ".join("%.1f" % x for x in row))
