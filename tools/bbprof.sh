#!/bin/bash
# dev: kernel stats of tools/${BBTOOL:-batchbbench.py} for the library named by $1 (old|new)
cd /tmp && export TMPDIR=/tmp
export DSA_DEV=1
if [ "$1" = "old" ]; then export DSA_LIBRARY=$GRAFT_REPO_ROOT/dynamicsparsearrays.jl_amd/csrc/libdsa_hip_old.so; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bbprof_$1 -o bb -- python3 $GRAFT_REPO_ROOT/tools/${BBTOOL:-batchbbench.py} > $GRAFT_REPO_ROOT/gpurun_out/bbprof_$1.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/bbprof_$1 -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -d, -f1-6
