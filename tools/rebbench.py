#!/usr/bin/env python3
"""Root-window rebalance micro-bench (dev tool): vectors of 2^20 .. 2^24 slots, HIP events on the launch stream.
Usage: python tools/rebbench.py [lg ...]"""
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
stream = torch.cuda.current_stream()
lgs = [int(a) for a in sys.argv[1:]] or [20, 21, 22, 24]
for lg in lgs:
    for dens in (0.35, 0.70):
        capv = 1 << lg
        n = min(int(dens * capv) + (2 if dens < 0.5 else 0), int(0.7 * capv))
        vv = dsa.dynamicsparsevec(np.arange(1, n + 1, dtype=np.int64) * 3, bench.unit12(40 + lg, n), binding=hip)
        assert vv.info()["capacity"] == capv
        hip.call("vec_set_stream", vv.h, C.c_void_p(stream.cuda_stream))
        line = "2^%d d=%.2f:" % (lg, dens)
        for label, fn in (("uniform", lambda: vv.rebalance_root()),):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nrep = 50 if lg < 24 else 20
            r0.record(stream)
            for _ in range(nrep):
                fn()
            r1.record(stream)
            torch.cuda.synchronize()
            us = r0.elapsed_time(r1) / nrep * 1e3
            line += "  %s %.1f us (%.3f alg, %.3f physical 12 B)" % (label, us, 32 * capv / us / 8e6, 24.25 * capv / us / 8e6)
        if True:
            for mode, label in ((1, "packed-left"), (2, "packed-right")):
                def both():
                    hip.call("vec_dev_relayout", vv.h, mode)
                    vv.rebalance_root()
                def one():
                    hip.call("vec_dev_relayout", vv.h, mode)
                ts = []
                for fn in (both, one):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    nrep = 30 if lg < 24 else 10
                    r0.record(stream)
                    for _ in range(nrep):
                        fn()
                    r1.record(stream)
                    torch.cuda.synchronize()
                    ts.append(r0.elapsed_time(r1) / nrep * 1e3)
                us = ts[0] - ts[1]
                line += "  %s->spread %.1f us (%.3f alg; relayout alone %.1f)" % (label, us, 32 * capv / us / 8e6, ts[1])
                vv.rebalance_root()
        print(line, flush=True)
        del vv
