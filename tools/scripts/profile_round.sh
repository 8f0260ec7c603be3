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
# the root rebalance alone (2^24 slots: every k_move2<false,...> row of this trace is that window), warm and cold; the SpMV cold
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt_reb -o kt --output-format csv -- python3 $R/tools/prof_rebalance.py warm 30 > $R/gpurun_out/${TAG}_kt_reb.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt_rebcold -o kt --output-format csv -- python3 $R/tools/prof_rebalance.py cold 12 > $R/gpurun_out/${TAG}_kt_rebcold.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt_spmvcold -o kt --output-format csv -- python3 $R/tools/prof_spmv.py cold > $R/gpurun_out/${TAG}_kt_spmvcold.log 2>&1
echo "rebalance-only / cold traces done"
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
for sub, name in (("kt_reb", "rebalance_only"), ("kt_rebcold", "rebalance_cold"), ("kt_spmvcold", "spmv_cold")):
    f = glob.glob("gpurun_out/${TAG}_%s/**/*kernel_stats.csv" % sub, recursive=True)
    if f: shutil.copy(f[0], "gpurun_out/${TAG}_%s_kernel_stats.csv" % name)
PY
cd /tmp
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
echo "bench done"
tail -c 2500 $R/gpurun_out/${TAG}_bench.json
