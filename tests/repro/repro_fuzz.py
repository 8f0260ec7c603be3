#!/usr/bin/env python3
"""Repro helper: one fuzz scenario (FUZZ_BIG honoured) with diagnostics where the layouts diverge.  usage: repro_fuzz.py <seed>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz

seed = int(sys.argv[1])
orig = fuzz.mat_equal
def mat_equal(a, b, ctx):
    try:
        orig(a, b, ctx)
    except AssertionError as e:
        print("DIVERGED at", ctx)
        for o in (0, 1):
            La, Lb = a.export_layout(o), b.export_layout(o)
            ia, ib = a.info(o), b.info(o)
            print(" orientation", o, {k: (ia[k], ib.get(k)) for k in ("capacity", "nb_elements", "stat_rebalances", "stat_window_slots", "stat_extends", "stat_par_rounds", "stat_par_ops", "stat_seq_ops") if k in ia})
            d = np.nonzero(La["occ"] != Lb["occ"])[0]
            print("  occ differs at %d slots" % len(d), "first", d[:10], "last", d[-5:] if len(d) else "")
            if len(d):
                lo, hi = int(d[0]), int(d[-1])
                print("  span", lo, hi, "width", hi - lo + 1, "popcount hip/ora in span", int(La["occ"][lo:hi + 1].sum()), int(Lb["occ"][lo:hi + 1].sum()))
                seg = La["info"]["segment_capacity"]
                w = 1
                while w < hi - lo + 1 or (lo // w) != (hi // w): w *= 2
                print("  smallest aligned window covering the span:", w, "slots (segment", seg, ") start", (lo // w) * w)
        raise
fuzz.mat_equal = mat_equal
print("result:", fuzz.run_matrix(seed) if seed % 4 else fuzz.run_vector(seed))
