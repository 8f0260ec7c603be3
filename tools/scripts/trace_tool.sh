# usage (on the GPU box, from the repo root): bash tools/scripts/trace_tool.sh <tag> <tool.py> [args]
# rocprofv3 kernel trace of one of the dev tools under tools/; the 30 heaviest kernels go to gpurun_out/<tag>_kernel_stats.txt
set -euo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 $R/tools/$@ > $R/gpurun_out/${TAG}_kt.log 2>&1
cd $R && python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/${TAG}_kt/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    with open("gpurun_out/${TAG}_kernel_stats.txt", "w") as out:
        for r in rows[:30]:
            out.write("%-80s calls %6s avg %9.1f us total %10.1f us\n" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
head -12 $R/gpurun_out/${TAG}_kernel_stats.txt
tail -4 $R/gpurun_out/${TAG}_kt.log
