#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/c5_split; mkdir -p $O
DSA_DEV=1 DSA_DBG_TIME=1 timeout -k 10 300 python tools/c5bench.py --full > $O/c5_time.log 2>&1; echo "rc=$?"
DSA_DEV=1 DSA_DBG_TIME=1 DSA_DBG_SPLIT=1 C5_BATCHES=12 timeout -k 10 300 python tools/c5bench.py --full > $O/c5_split12.log 2>&1; echo "rc=$?"
timeout -k 10 300 python tools/c5bench.py --full > $O/c5_plain.log 2>&1; tail -3 $O/c5_plain.log
