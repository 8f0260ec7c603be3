# usage (on the GPU box, from the repo root): bash tools/scripts/trace_fill.sh <tag>
# K-build / closefillmode! timing (tools/fillbench.py, DSA_DBG_TIME split) and a rocprofv3 kernel trace of the same program
set -euo pipefail
TAG=${1:-fill}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
DSA_DEV=1 DSA_DBG_TIME=1 python3 $R/tools/fillbench.py > $R/gpurun_out/${TAG}_fillbench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 $R/tools/fillbench.py > $R/gpurun_out/${TAG}_kt.log 2>&1
cd $R && python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/${TAG}_kt/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    with open("gpurun_out/${TAG}_kernel_stats.txt", "w") as out:
        for r in rows[:40]:
            out.write("%-70s calls %6s avg %10.1f us total %10.1f us\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
cat $R/gpurun_out/${TAG}_fillbench.log | grep -v "^\[mat_apply" | tail -40
cat $R/gpurun_out/${TAG}_kernel_stats.txt | head -30
