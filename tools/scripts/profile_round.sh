# usage (on the GPU box, from the repo root): bash tools/scripts/profile_round.sh <tag>
# separate PMC passes (FETCH_SIZE / WRITE_SIZE), rocprofv3 kernel traces of bench.py, then the plain bench line (which quotes the
# fresh PMC traffic); everything lands in gpurun_out/<tag>_* (+ profiles/<tag>_* written by summarize_prof.py)
set -euo pipefail
TAG=${1:-r01_final}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc_fetch -o f --output-format csv -- python3 $R/tools/prof_kernels.py > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc_write -o w --output-format csv -- python3 $R/tools/prof_kernels.py > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
echo "pmc done"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt_spmv -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_spmv_only_bench.json 2> $R/gpurun_out/${TAG}_kt_spmv.err
echo "spmv-only trace done"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_kt_bench.json 2> $R/gpurun_out/${TAG}_kt.err
echo "full trace done"
cd $R
python3 tools/summarize_prof.py $TAG gpurun_out/${TAG}_kt gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write > gpurun_out/${TAG}_summary.log 2>&1
cp profiles/${TAG}_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
cp profiles/${TAG}_pmc_summary.json gpurun_out/${TAG}_pmc_summary.json
python3 - <<PY
import glob, shutil
f = glob.glob("gpurun_out/${TAG}_kt_spmv/**/*kernel_stats.csv", recursive=True)
if f: shutil.copy(f[0], "gpurun_out/${TAG}_spmv_only_kernel_stats.csv")
PY
cd /tmp
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
echo "bench done"
tail -c 2500 $R/gpurun_out/${TAG}_bench.json
