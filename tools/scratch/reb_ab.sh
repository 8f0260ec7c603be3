# A/B of k_move2 build variants on ONE box (box-to-box spread is ~5 %): rebuilds rebalance.o with each -D set and runs tools/rebbench.py
set -uo pipefail
cd ${GRAFT_REPO_ROOT:?}
for V in "" "-DM2_NO_INNER" "" "-DM2_NO_INNER"; do
  touch dynamicsparsearrays.jl_amd/csrc/rebalance.hip
  make -C dynamicsparsearrays.jl_amd/csrc -j8 EXTRA="$V" > /dev/null 2>&1
  echo "== variant [$V]"
  python tools/rebbench.py 20 21 22 24 2>&1 | grep "2^" | cut -c1-75
done
touch dynamicsparsearrays.jl_amd/csrc/rebalance.hip; make -C dynamicsparsearrays.jl_amd/csrc -j8 > /dev/null 2>&1
python -m pytest tests -m gpu -x -q -k "rebalance or relayout or raw or spread or bulk or c3_full or extend" 2>&1 | tail -2
