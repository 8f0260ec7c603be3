# ablation: k_apply without the per-op atomicAdd on Ctl::nb_elements (counts go wrong: timing only)
set -uo pipefail
cd ${GRAFT_REPO_ROOT:?}
for V in "" "-DPB_ABL_NOCOUNT" ""; do
  touch dynamicsparsearrays.jl_amd/csrc/parbatch.hip
  make -C dynamicsparsearrays.jl_amd/csrc -j8 EXTRA="$V" > /dev/null 2>&1
  echo "== variant [$V]"
  python tools/batchbbench.py 2>&1 | grep -E "batch B|matrix random"
done
