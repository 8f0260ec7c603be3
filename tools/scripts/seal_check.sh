# check of a resolver change: early config-5 batches and batch B timed, the GPU suite, then the footprint-check build (both modes) and the product library on the fuzzer
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:-resolver_check}; SEC=${2:-120}; SEED=${3:-91000}
O=$R/gpurun_out/$TAG
mkdir -p $O
python tools/c5early.py 2>/dev/null | grep "^batch" > $O/c5early.log; awk '{s+=$3; printf "%s ", $3} END {print " | sum " s}' $O/c5early.log
python tools/batchbbench.py 2>/dev/null > $O/bb.log; cut -c1-120 $O/bb.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
bash tools/scripts/fpcheck_soak.sh $TAG $SEC 1 $SEED || exit 1
bash tools/scripts/fpcheck_soak.sh $TAG $SEC 2 $((SEED + 1000)) || exit 1
FUZZ_ONLY=leafmat timeout -k 10 $((SEC + 120)) python tools/fuzz.py $SEC $((SEED + 2000)) > $O/product_leafmat.log 2>&1; tail -1 $O/product_leafmat.log
timeout -k 10 $((SEC + 120)) python tools/fuzz.py $SEC $((SEED + 3000)) > $O/product_mix.log 2>&1; tail -1 $O/product_mix.log
