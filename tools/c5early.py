import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import bench, dsa_loader
dsa = dsa_loader.load(); hip = dsa.product()
m5, ncols5, per5, every = bench.C5_FULL
I5, J5, V5 = bench.c5_columns(m5, ncols5, per5)
W = dsa.dynamicsparse(fill_mode=False, binding=hip)
W.set_batch(I5[:16000], J5[:16000], V5[:16000]); del W
B = dsa.dynamicsparse(fill_mode=False, binding=hip)
keys = ("capacity", "nb_elements", "stat_rebalances", "stat_extends", "stat_par_rounds", "stat_par_ops", "stat_seq_ops")
prev = [dict.fromkeys(keys, 0), dict.fromkeys(keys, 0)]
for b in range(14):
    sl = slice(b * every * per5, (b + 1) * every * per5)
    t = time.perf_counter(); B.set_batch(I5[sl], J5[sl], V5[sl]); hip.call("mat_sync", B.h); dt = time.perf_counter() - t
    out = []
    for o in (0, 1):
        inf = B.info(o)
        d = {k: inf[k] - prev[o][k] for k in keys[2:]}
        prev[o] = {k: inf[k] for k in keys}
        out.append("cap %d n %d reb %d ext %d rounds %d par %d seq %d" % (inf["capacity"], inf["nb_elements"], d["stat_rebalances"], d["stat_extends"], d["stat_par_rounds"], d["stat_par_ops"], d["stat_seq_ops"]))
    print("batch %2d %.2f ms | col: %s | row: %s" % (b, dt * 1e3, out[0], out[1]), flush=True)
    lib = getattr(hip, "lib", None)      # a -DDSA_PB_PROF build (DSA_LIBRARY): the resolve step's phases of this batch, to stderr
    if lib is not None and hasattr(lib, "dsa_dbg_pbprof_dump"):
        sys.stderr.flush(); lib.dsa_dbg_pbprof_dump()
