#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
C=$R/dynamicsparsearrays.jl_amd/csrc
O=$R/gpurun_out/nofb; mkdir -p $O
DSA_DEV=1 DSA_DBG_TIME=1 DSA_DBG_SPLIT=1 timeout -k 10 300 python tools/c5bench.py --full > $O/c5_time.log 2>&1; echo "rc=$?"; tail -3 $O/c5_time.log | head -1
grep "mat_apply_sets" $O/c5_time.log | awk 'NR>1{tot+=$5; n++; if(NR<=11){f10+=$5}} END{print n, "batches total", tot, "ms; first10", f10}'
for m in 1 2; do
DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$m timeout -k 10 500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "c5 or column or streaming or newcol or parallel" > $O/tests_mode$m.log 2>&1; echo "mode $m tests rc=$?"; tail -2 $O/tests_mode$m.log
DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$m timeout -k 10 300 python tools/fuzz.py 90 $((8100 + m)) > $O/fuzz_mode$m.log 2>&1; echo "mode $m fuzz rc=$?"; tail -1 $O/fuzz_mode$m.log
DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$m FUZZ_ONLY=leafmat timeout -k 10 300 python tools/fuzz.py 60 $((9100 + m)) > $O/fuzz_leafmat_mode$m.log 2>&1; echo "mode $m fuzz leafmat rc=$?"; tail -1 $O/fuzz_leafmat_mode$m.log
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
