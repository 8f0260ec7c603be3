#!/bin/bash
# first run of the footprint-check build on the GPU: known regressions on the _bugN libraries, then the suite's write tests and the fuzzer
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
C=$R/dynamicsparsearrays.jl_amd/csrc
O=$R/gpurun_out/fp_first
mkdir -p $O
echo "== bug1 lib, mode 1 (expect DSA_FP_CHECK failure)"
DSA_LIBRARY=$C/libdsa_hip_fpcheck_bug1.so DSA_FP_MODE=1 timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k two_deletes_from_a_leaf > $O/bug1_mode1.log 2>&1; echo "rc=$?"; grep -c DSA_FP_CHECK $O/bug1_mode1.log
echo "== bug1 lib, mode 2"
DSA_LIBRARY=$C/libdsa_hip_fpcheck_bug1.so DSA_FP_MODE=2 timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k two_deletes_from_a_leaf > $O/bug1_mode2.log 2>&1; echo "rc=$?"; grep -c DSA_FP_CHECK $O/bug1_mode2.log
echo "== bug2 lib, mode 1, fuzz leaf seeds 1990.."
DSA_LIBRARY=$C/libdsa_hip_fpcheck_bug2.so DSA_FP_MODE=1 FUZZ_ONLY=leaf timeout -k 10 300 python tools/fuzz.py 8 1990 > $O/bug2_mode1.log 2>&1; echo "rc=$?"; grep -c DSA_FP_CHECK $O/bug2_mode1.log
echo "== bug2 lib, mode 2"
DSA_LIBRARY=$C/libdsa_hip_fpcheck_bug2.so DSA_FP_MODE=2 FUZZ_ONLY=leaf timeout -k 10 300 python tools/fuzz.py 8 1990 > $O/bug2_mode2.log 2>&1; echo "rc=$?"; grep -c DSA_FP_CHECK $O/bug2_mode2.log
echo "== check lib on the fixed sources: the same two scenarios must be clean"
for m in 1 2; do
DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$m timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k two_deletes_from_a_leaf > $O/ok_leaf_mode$m.log 2>&1; echo "mode $m leaf test rc=$?"
DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$m FUZZ_ONLY=leaf timeout -k 10 300 python tools/fuzz.py 8 1990 > $O/ok_fuzzleaf_mode$m.log 2>&1; echo "mode $m fuzz leaf rc=$?"; tail -1 $O/ok_fuzzleaf_mode$m.log
DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$m timeout -k 10 400 python tools/fuzz.py 60 $((4000 + m)) > $O/ok_fuzz_mode$m.log 2>&1; echo "mode $m fuzz mix rc=$?"; tail -1 $O/ok_fuzz_mode$m.log
done
echo "== whole gpu suite on the check lib, mode 1"
DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=1 timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/suite_mode1.log 2>&1; echo "rc=$?"; tail -3 $O/suite_mode1.log
