#!/bin/bash
# after a change of the plan / resolve / apply kernels: check build (both modes) on the targeted scenarios + default mix, the suite, the insert legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
C=$R/dynamicsparsearrays.jl_amd/csrc
TAG=${1:-plan}; SEC=${2:-60}
O=$R/gpurun_out/$TAG; mkdir -p $O
rc=0
for m in 1 2; do
  for only in leaf leafmat tomb; do
    DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$m FUZZ_ONLY=$only timeout -k 10 300 python tools/fuzz.py $((SEC / 2)) $((11000 + m)) > $O/fuzz_${only}_mode$m.log 2>&1 || rc=1
    echo "mode $m $only: $(tail -1 $O/fuzz_${only}_mode$m.log)"
  done
  DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$m timeout -k 10 400 python tools/fuzz.py $SEC $((12000 + m)) > $O/fuzz_mix_mode$m.log 2>&1 || rc=1
  echo "mode $m mix: $(tail -1 $O/fuzz_mix_mode$m.log)"
done
grep -l DSA_FP_CHECK $O/*.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1 || rc=1; tail -2 $O/suite.log
python tools/seqbench.py > $O/seqbench.log 2>&1; tail -4 $O/seqbench.log
python tools/batchbbench.py > $O/batchb.log 2>&1; tail -3 $O/batchb.log
timeout -k 10 300 python tools/c5bench.py --full > $O/c5.log 2>&1; tail -3 $O/c5.log | head -1
exit $rc
