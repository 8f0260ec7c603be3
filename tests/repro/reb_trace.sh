# kernel durations of the root rebalance at 2^20 / 2^21 (rocprofv3 trace) beside the HIP-event timing of tools/rebbench.py
set -euo pipefail
R=${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/rebtrace -o rt --output-format csv -- python3 $R/tools/rebbench.py 20 21 > $R/gpurun_out/rebtrace.log 2>&1
cd $R && python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/rebtrace/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_move2" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(d.items()):
        v.sort()
        print(k, "n", len(v), "median %.2f us" % (v[len(v)//2] / 1e3), "min %.2f" % (v[0] / 1e3), "p90 %.2f" % (v[int(len(v)*0.9)] / 1e3))
PY
grep "2^" gpurun_out/rebtrace.log
