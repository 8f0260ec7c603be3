#!/usr/bin/env python3
"""Summarises rocprofv3 output directories into profiles/: usage
   summarize_prof.py <tag> <kernel-trace dir> <fetch dir> <write dir>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, kt, fetch, write = sys.argv[1:5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
stats = glob.glob(os.path.join(kt, "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out_dir, f"{tag}_kernel_stats.csv"))


def per_kernel(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg


F, W = per_kernel(fetch), per_kernel(write)
KB = 1024.0
res = {"_units": "bytes per dispatch; rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB",
       "_correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a coalesced stream -> x2 on the stream part "
                      "(MI355X_MICROARCH.md §HBM); calibrated here on k_spmv_gather with nx=0 (streams only) and on k_move"}
sp = [k for k in F if "k_spmv_gather" in k]
if sp:
    f = F[sp[0]]
    w = W[sp[0]]
    full, stream = sum(f[0:3]) / 3 * KB, sum(f[3:6]) / 3 * KB
    res["k_spmv_gather_C3"] = {
        "FETCH_SIZE_full": round(full), "FETCH_SIZE_streams_only_nx0": round(stream),
        "WRITE_SIZE_full": round(sum(w[0:3]) / 3 * KB),
        "stream_bytes_expected": 12 * 16777216 + 2097152,       # physical stream: int32 keys + Float64 values + the occupancy bitmap
        "stream_calibration_factor": round((12 * 16777216 + 2097152) / stream, 3),
        "corrected_fetch": round(2 * stream + (full - stream)),
        "corrected_traffic_total": round(2 * stream + (full - stream) + sum(w[0:3]) / 3 * KB),
        "algorithmic_bytes": 16 * 16777216 + 16 * 1000000,
    }
mv = [k for k in F if "k_move2<false" in k] or [k for k in F if "k_move<false" in k]
if mv:
    f, w = F[mv[0]][-3:], W[mv[0]][-3:]
    res["k_move_root_2^24"] = {"FETCH_SIZE": round(sum(f) / 3 * KB), "WRITE_SIZE": round(sum(w) / 3 * KB),
                               "corrected_fetch": round(2 * sum(f) / 3 * KB),
                               "corrected_traffic_total": round(2 * sum(f) / 3 * KB + sum(w) / 3 * KB),
                               "algorithmic_bytes": 32 * 16777216}
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha: bench.py quotes this file only while the kernel sources are unchanged)
res["kernel_source_sha"] = bench.kernel_source_sha()
json.dump(res, open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
