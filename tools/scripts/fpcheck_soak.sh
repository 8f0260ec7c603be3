#!/bin/bash
# soak of the footprint-check build (csrc/parbatch.hip, -DDSA_FP_CHECK): tools/fuzz.py on libdsa_hip_fpcheck.so
#   usage: fpcheck_soak.sh <tag> <seconds per leg> <mode> <first seed>      legs: default mix, FUZZ_BIG, default mix with 64-bit keys
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=${1:-soak}; SEC=${2:-300}; MODE=${3:-1}; SEED=${4:-70000}
C=$R/dynamicsparsearrays.jl_amd/csrc
O=$R/gpurun_out/$TAG
mkdir -p $O
export DSA_DEV=1 DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=$MODE
rc=0
timeout -k 10 $((SEC + 120)) python tools/fuzz.py $SEC $SEED > $O/mix_mode$MODE.log 2>&1 || rc=1
tail -1 $O/mix_mode$MODE.log
FUZZ_BIG=1 timeout -k 10 $((SEC + 200)) python tools/fuzz.py $SEC $((SEED + 100000)) > $O/big_mode$MODE.log 2>&1 || rc=1
tail -1 $O/big_mode$MODE.log
DSA_KEYS_WIDE=1 timeout -k 10 $((SEC / 2 + 120)) python tools/fuzz.py $((SEC / 2)) $((SEED + 200000)) > $O/mix_wide_mode$MODE.log 2>&1 || rc=1
tail -1 $O/mix_wide_mode$MODE.log
grep -c DSA_FP_CHECK $O/*.log
exit $rc
