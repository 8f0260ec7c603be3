# soak of the one-pass sealing (round 6): footprint-check build in shadow mode, then the product library on the leaf-matrix and big scenarios
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/scripts/fpcheck_soak.sh seal_soak2 150 2 83000
O=$R/gpurun_out/seal_soak2
FUZZ_ONLY=leafmat timeout -k 10 260 python tools/fuzz.py 150 84000 > $O/product_leafmat.log 2>&1; tail -1 $O/product_leafmat.log
FUZZ_BIG=1 timeout -k 10 320 python tools/fuzz.py 150 85000 > $O/product_big.log 2>&1; tail -1 $O/product_big.log
timeout -k 10 260 python tools/fuzz.py 150 86000 > $O/product_mix.log 2>&1; tail -1 $O/product_mix.log
