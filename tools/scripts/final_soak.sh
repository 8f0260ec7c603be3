# long soak of the final libraries: footprint-check build (mode given) on the mix / big / wide legs, then the same-leaf scenarios on it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
MODE=${1:-1}; SEC=${2:-240}; SEED=${3:-110000}
TAG=final_soak_m$MODE
bash tools/scripts/fpcheck_soak.sh $TAG $SEC $MODE $SEED || exit 1
O=$R/gpurun_out/$TAG
export DSA_DEV=1 DSA_LIBRARY=$R/dynamicsparsearrays.jl_amd/csrc/libdsa_hip_fpcheck.so DSA_FP_MODE=$MODE
FUZZ_ONLY=leaf timeout -k 10 $((SEC / 2 + 120)) python tools/fuzz.py $((SEC / 2)) $((SEED + 300000)) > $O/leaf_mode$MODE.log 2>&1 || exit 1
tail -1 $O/leaf_mode$MODE.log
FUZZ_ONLY=leafmat timeout -k 10 $((SEC / 2 + 120)) python tools/fuzz.py $((SEC / 2)) $((SEED + 400000)) > $O/leafmat_mode$MODE.log 2>&1 || exit 1
tail -1 $O/leafmat_mode$MODE.log
grep -c DSA_FP_CHECK $O/leaf_mode$MODE.log $O/leafmat_mode$MODE.log || true      # (0 matches is the good case)
