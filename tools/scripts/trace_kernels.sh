set -euo pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt --output-format csv -- python3 $R/tools/prof_kernels.py > $R/gpurun_out/kt.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsa::" in r["Name"]:
            print(r["Name"][:50], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
