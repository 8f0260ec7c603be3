#!/usr/bin/env python3
"""Launch sequence for a kernel trace of the ROOT REBALANCE ALONE (run under rocprofv3 --kernel-trace --stats): the C3 matrix is
built, then nothing but 2^24-slot root rebalances of its colmajor orientation are launched — every k_move2<false, ...> row of the
trace is that window, so `roofline_rebalance` of the bench line can be recomputed from the kernel-stats CSV alone (the builds use
the PACKED instantiation).  `cold`: a 1 GiB device write (a fill kernel) in front of every launch, as bench.py's cold_* numbers.

    python tools/prof_rebalance.py [warm|cold] [launches]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dsa_loader  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "warm"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dsa = dsa_loader.load()
hip = dsa.product()
m = n = 1_000_000
I, J, V = bench.c3_triplets(m, n, 10, 0, 5, 6)
A = dsa.dynamicsparse(I, J, V, m, n, binding=hip)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
hip.call("mat_set_stream", A.h, C.c_void_p(stream.cuda_stream))
cap = A.info(dsa.COLMAJOR)["capacity"]
scr = torch.empty((1 << 30) // 4, dtype=torch.float32, device=dev) if mode == "cold" else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for i in range(launches):
    if scr is not None:
        scr.fill_(float(i))
    e0.record(stream)
    A.rebalance_root(dsa.COLMAJOR)
    e1.record(stream)
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("root rebalance of %d slots, %s, %d launches: median %.2f us (HIP events), algorithmic %d B -> %.3f of 8 TB/s" %
      (cap, mode, launches, ts[len(ts) // 2], 32 * cap, 32 * cap / ts[len(ts) // 2] / 1e3 / 8000.0))
