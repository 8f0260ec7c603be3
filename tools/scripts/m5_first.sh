#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/m5_first; mkdir -p $O
FUZZ_ONLY=append timeout -k 10 200 python tools/fuzz.py 60 5000 > $O/fuzz_append.log 2>&1; echo "fuzz append rc=$?"; tail -2 $O/fuzz_append.log
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "c5 or append or column_generation or streaming" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
DSA_DEV=1 DSA_DBG_RUN=1 timeout -k 10 300 python tools/c5bench.py --full > $O/c5bench.log 2>&1; echo "c5bench rc=$?"; grep -v "previous append run\|model v2" $O/c5bench.log | tail -12; grep "append model" $O/c5bench.log | tail -4
DSA_DEV=1 DSA_MODEL5=0 timeout -k 10 300 python tools/c5bench.py --full > $O/c5bench_m5off.log 2>&1; tail -4 $O/c5bench_m5off.log
