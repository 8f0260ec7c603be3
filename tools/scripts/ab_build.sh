# usage (on the GPU box, from the repo root): bash tools/scripts/ab_build.sh <source.hip> "<EXTRA flags A>" "<EXTRA flags B>" -- <command ...>
# A/B of two builds of one translation unit on ONE box (box-to-box spread is 3-5 %, more than most kernel changes): rebuilds the
# unit with each flag set (twice each, A B A B) and runs the command after every build.  Example:
#   bash tools/scripts/ab_build.sh rebalance.hip "" "-DM2_SOME_VARIANT" -- python tools/rebbench.py 20 21 24
set -uo pipefail
cd ${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
SRC=$1; A=$2; B=$3; shift 3; [ "$1" = "--" ] && shift
C=dynamicsparsearrays.jl_amd/csrc
for V in "$A" "$B" "$A" "$B"; do
  touch $C/$SRC
  make -C $C -j8 EXTRA="$V" > /dev/null 2>&1 || { echo "build failed with [$V]"; exit 1; }
  echo "== EXTRA=[$V]"
  "$@"
done
touch $C/$SRC; make -C $C -j8 > /dev/null 2>&1
