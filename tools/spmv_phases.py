#!/usr/bin/env python3
"""Phase stamps of k_spmv_gather (dev tool; needs a library built with -DSPMV_PROF, tools/scripts/spmv_variants.sh, selected through
DSA_LIBRARY): runs the product on the banded / config-5 shapes of tools/spmv_ab.py and prints, per phase, the distribution of the
per-wave durations in shader clocks, and how the waves' lifetimes overlap per CU.   usage: spmv_phases.py banded|c5|c3"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dsa_loader  # noqa: E402

dsa = dsa_loader.load()
hip = dsa.product()
lib = C.CDLL(os.environ["DSA_LIBRARY"])
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "banded"
if which == "c5":
    m5, ncols5, per5, every = bench.C5_FULL
    I5, J5, V5 = bench.c5_columns(m5, ncols5, per5)
    A = dsa.dynamicsparse(fill_mode=False, binding=hip)
    for c0 in range(0, ncols5, every):
        sl = slice(c0 * per5, (c0 + every) * per5)
        A.set_batch(I5[sl], J5[sl], V5[sl])
    hip.call("mat_sync", A.h)
    nx, ny, x = ncols5, m5, bench.unit12(13, ncols5)
elif which == "banded":
    mb = nb = 1_000_000
    z = bench.splitmix_array(51, nb * 10)
    colb = np.repeat(np.arange(1, nb + 1, dtype=np.int64), 10)
    rowb = np.clip(colb + (z % np.uint64(8192)).astype(np.int64) - 4096, 1, mb)
    keyb = colb * np.int64(mb + 1) + rowb
    _, firstb = np.unique(keyb, return_index=True)
    A = dsa.dynamicsparse(rowb[firstb], colb[firstb], bench.unit12(52, len(firstb)), mb, nb, binding=hip)
    nx, ny, x = nb, mb, bench.unit12(53, nb)
else:
    m = n = 1_000_000
    I, J, V = bench.c3_triplets(m, n, 10, 0, 5, 6)
    A = dsa.dynamicsparse(I, J, V, m, n, binding=hip)
    nx, ny, x = n, m, bench.unit12(7, n)
xd = torch.from_numpy(x).to(dev)
yd = torch.zeros(ny, dtype=torch.float64, device=dev)
for _ in range(4):
    hip.call("mat_spmv_dense_dev", A.h, 0, 0, C.c_void_p(xd.data_ptr()), nx, C.c_void_p(yd.data_ptr()), ny)
torch.cuda.synchronize()
cap = A.info(dsa.ROWMAJOR)["capacity"]
nw = min(cap // 512, 1 << 16)
buf = np.zeros(16 * nw, dtype=np.uint64)
assert lib.dsa_dbg_spmv_prof(buf.ctypes.data_as(C.c_void_p), C.c_long(16 * nw)) == 0
s = buf.reshape(nw, 16).astype(np.int64)
t0 = s[:, 0].min()
names = ["stream->masks+gather issue", "gathers arrive", "LDS phase + barrier", "walk", "row keys + stores drain"]
print("%s: capacity %d, %d waves stamped; kernel span (first start -> last end) %d clk" % (which, cap, nw, s[:, 5].max() - t0))
for i, nme in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print("  %-28s mean %7.0f  p10 %6d  p50 %6d  p90 %6d  max %7d" % (nme, d.mean(), *np.percentile(d, [10, 50, 90]).astype(int), d.max()))
life = s[:, 5] - s[:, 0]
print("  %-28s mean %7.0f  p10 %6d  p50 %6d  p90 %6d" % ("wave lifetime", life.mean(), *np.percentile(life, [10, 50, 90]).astype(int)))
print("  semaphores per span: mean %.1f" % s[:, 7].mean())
ok = s[:, 8] > 0
for nme, i0, i1 in (("walk: list + id reads", 3, 8), ("walk: row-key wait (vmcnt)", 8, 9), ("walk: reads + additions", 9, 10), ("walk: zero fill + store issue", 10, 11), ("walk: after the loop", 11, 4)):
    d = (s[ok, i1] - s[ok, i0])
    print("  %-30s mean %7.0f  p10 %6d  p50 %6d  p90 %6d" % (nme, d.mean(), *np.percentile(d, [10, 50, 90]).astype(int)))
hw = s[:, 6]
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4)       # CU_ID bits 11:8, SE_ID bits 15:13 (gfx9 HW_ID layout)
st = np.sort(s[:, 0] - t0)
print("  wave starts: p10 %d p50 %d p90 %d max %d ; distinct (se, cu) ids seen %d" % (*np.percentile(st, [10, 50, 90]).astype(int), st.max(), len(np.unique(cu))))
# how many waves are inside each phase at sampled instants (whole chip)
T = np.linspace(0, s[:, 5].max() - t0, 9)[1:-1]
for tt in T:
    inside = [(int(((s[:, i] - t0) <= tt).sum() - ((s[:, i + 1] - t0) <= tt).sum())) for i in range(5)]
    print("  t=%7d clk: waves in phase %s" % (tt, inside))
