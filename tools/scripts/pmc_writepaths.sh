# usage (GPU box, repo root): bash tools/scripts/pmc_writepaths.sh
# SQ counters of the write-path kernels (k_append_run: one wave; k_plan / k_apply: one wave per op) on tools/appendbench.py and
# tools/batchbbench.py -> gpurun_out/writepath_sq_counters.json
set -euo pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES -d $R/gpurun_out/pmc_wp_a -o a --output-format csv -- python3 $R/tools/appendbench.py > $R/gpurun_out/pmc_wp_a.log 2>&1 && \
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES -d $R/gpurun_out/pmc_wp_b -o b --output-format csv -- python3 $R/tools/batchbbench.py > $R/gpurun_out/pmc_wp_b.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d, kernels in (("gpurun_out/pmc_wp_a", ("k_append_run",)), ("gpurun_out/pmc_wp_b", ("k_plan", "k_apply"))):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    for kn in kernels:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in rows:
            if kn in r["Kernel_Name"]:
                agg[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        res = {}
        for c, dd in agg.items():
            v = [dd[k] for k in sorted(dd)]
            res[c] = {"dispatches": len(v), "avg_per_dispatch": round(sum(v) / max(len(v), 1), 1), "max": round(max(v), 1)}
        wc = res.get("SQ_WAVE_CYCLES", {}).get("avg_per_dispatch"); wi = res.get("SQ_WAIT_INST_ANY", {}).get("avg_per_dispatch")
        ins = sum(res.get(k, {}).get("avg_per_dispatch", 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
        if wc:
            res["_derived"] = {"wait_inst_any_share_of_wave_cycles": round(wi / wc, 4) if wi else None,
                               "wave_cycles_per_counted_instruction": round(wc / ins, 2) if ins else None}
        out[kn] = res
json.dump({"note": "rocprofv3 --pmc (one pass per tool), sums over all XCDs / SEs per dispatch; k_append_run on tools/appendbench.py "
                   "(two reps of config 2 batch A), k_plan / k_apply on tools/batchbbench.py (config 2 batch B x2 + 200 k matrix updates)",
           "counters": out}, open("gpurun_out/writepath_sq_counters.json", "w"), indent=1)
print(json.dumps({k: v.get("_derived") for k, v in out.items()}, indent=1))
PY
