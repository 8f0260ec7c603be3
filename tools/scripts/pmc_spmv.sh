# usage (GPU box, repo root): bash tools/scripts/pmc_spmv.sh [tag]   -> gpurun_out/<tag>_spmv_sq_tcc_counters.json
set -euo pipefail
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- python3 $R/tools/prof_spmv.py > $R/gpurun_out/pmc_sq.log 2>&1 && \
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/pmc_tc -o tc --output-format csv -- python3 $R/tools/prof_spmv.py > $R/gpurun_out/pmc_tc.log 2>&1
cd $R && TAG=$TAG python3 - <<'PY'
# per-dispatch sums of every counter for the k_spmv_gather launches: dispatches 0-2 = C3 product, 3-5 = the same with nx = 0 (stream only)
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
out = {}
for d in ("gpurun_out/pmc_sq", "gpurun_out/pmc_tc"):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if "k_spmv_gather" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, dd in agg.items():
        v = [dd[k] for k in sorted(dd)]
        out[c] = {"C3_product_avg": round(sum(v[:3]) / 3), "stream_only_nx0_avg": round(sum(v[3:6]) / 3) if len(v) >= 6 else None}
h, m = out.get("TCC_HIT_sum", {}).get("C3_product_avg"), out.get("TCC_MISS_sum", {}).get("C3_product_avg")
if h is not None and m is not None:
    out["l2_hit_rate_C3_product"] = round(h / (h + m), 4)
w, wi = out.get("SQ_WAVE_CYCLES", {}).get("C3_product_avg"), out.get("SQ_WAIT_INST_ANY", {}).get("C3_product_avg")
if w and wi:
    out["wait_inst_any_share_of_wave_cycles"] = round(wi / w, 4)
json.dump({"note": "rocprofv3 --pmc, two separate passes (SQ_*, TCC/TCP), k_spmv_gather on the C3 matrix; sums over all XCDs / SEs per dispatch",
           "kernel_source_sha": bench.kernel_source_sha(),
           "counters": out}, open("gpurun_out/%s_spmv_sq_tcc_counters.json" % os.environ.get("TAG", "r04"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
