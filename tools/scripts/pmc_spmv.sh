set -euo pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- python3 $R/tools/prof_spmv.py > $R/gpurun_out/pmc_sq.log 2>&1 && \
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/pmc_tc -o tc --output-format csv -- python3 $R/tools/prof_spmv.py > $R/gpurun_out/pmc_tc.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_sq", "gpurun_out/pmc_tc"):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "spmv" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    for c, dd in agg.items():
        print(c, [round(sum(v)) for k, v in sorted(dd.items())])
PY
