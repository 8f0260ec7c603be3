set -uo pipefail
cd ${GRAFT_REPO_ROOT:?}
touch dynamicsparsearrays.jl_amd/csrc/parbatch.hip
make -C dynamicsparsearrays.jl_amd/csrc -j8 EXTRA="-DDSA_PB_PROF" > /dev/null 2>&1
python tools/batchbbench.py > gpurun_out/plan_prof.log 2>&1
grep -c resolve gpurun_out/plan_prof.log
