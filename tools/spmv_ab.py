#!/usr/bin/env python3
"""A/B of the gather kernel (dev tool): time y = A x on the three shapes bench.py reports — config 3 (uniformly random columns, x = 8 MB),
the banded extra (every gather an L2 hit), the final config-5 matrix (2^21 slots, x = 400 KB) — with HIP events around N launches, and
print y's bytes as a sha256 so that two variants of the kernel (run this tool once per library — DSA_DEV=1 DSA_LIBRARY=<another build of csrc/> —
or per setting of the development switches, e.g. DSA_DEV=1 DSA_SPMV_SHARE=0) can be compared bit for bit (the banded shape has two
rows longer than a span: their sums are joined by atomics and may differ in the last bits from run to run).  usage: python3 tools/spmv_ab.py [c3] [banded] [c4] [c5]"""
import ctypes as C
import hashlib
import json
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
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["c3", "banded", "c5"]
out = {"switches": {k: v for k, v in os.environ.items() if k.startswith("DSA_")}}


def run(name, A, nx, ny, x, reps=40):
    hip.call("mat_set_stream", A.h, C.c_void_p(stream.cuda_stream))
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros(ny, dtype=torch.float64, device=dev)
    fn = lambda: hip.call("mat_spmv_dense_dev", A.h, 0, 0, C.c_void_p(xd.data_ptr()), nx, C.c_void_p(yd.data_ptr()), ny)  # noqa: E731
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / reps)
    y = yd.cpu().numpy()
    out[name] = {"us_median": round(float(np.median(best)), 2), "us_min": round(float(min(best)), 2), "capacity": A.info(dsa.ROWMAJOR)["capacity"],
                 "y_sha256": hashlib.sha256(y.tobytes()).hexdigest()[:16], "y_sum": float(y.sum())}


if "c3" in which:
    m = n = 1_000_000
    I, J, V = bench.c3_triplets(m, n, 10, 0, 5, 6)
    A = dsa.dynamicsparse(I, J, V, m, n, binding=hip)
    run("c3", A, n, m, bench.unit12(7, n))
    del A
if "banded" in which:
    mb = nb = 1_000_000
    z = bench.splitmix_array(51, nb * 10)
    colb = np.repeat(np.arange(1, nb + 1, dtype=np.int64), 10)
    rowb = np.clip(colb + (z % np.uint64(8192)).astype(np.int64) - 4096, 1, mb)
    keyb = colb * np.int64(mb + 1) + rowb
    _, firstb = np.unique(keyb, return_index=True)
    A = dsa.dynamicsparse(rowb[firstb], colb[firstb], bench.unit12(52, len(firstb)), mb, nb, binding=hip)
    run("banded", A, nb, mb, bench.unit12(53, nb))
    del A
if "c4" in which:
    m4, n4 = 10_000_000, 1_250_000
    I4, J4, V4 = bench.c3_triplets(m4, n4, 10, 0, seed_rows=8, seed_vals=9)
    A = dsa.dynamicsparse(I4, J4, V4, m4, n4, binding=hip)
    run("c4_shard", A, n4, m4, bench.unit12(10, n4), reps=20)
    del A
if "c5" in which:
    m5, ncols5, per5, every = bench.C5_FULL
    I5, J5, V5 = bench.c5_columns(m5, ncols5, per5)
    B = dsa.dynamicsparse(fill_mode=False, binding=hip)
    for c0 in range(0, ncols5, every):
        sl = slice(c0 * per5, (c0 + every) * per5)
        B.set_batch(I5[sl], J5[sl], V5[sl])
    hip.call("mat_sync", B.h)
    run("c5_final", B, ncols5, m5, bench.unit12(13, ncols5), reps=200)
    del B
print(json.dumps(out))
