# builds libdsa_hip_<tag>.so: csrc/spmv.hip with the given -D switches, everything else from the objects of the product build (run
# `make -C csrc` first).  Select one with DSA_LIBRARY=<path> (binding.py).   usage: spmv_variants.sh tag:"-DA=1 -DB=2" ...
set -euo pipefail
cd "$(dirname "$0")/../../dynamicsparsearrays.jl_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-value"
OBJS="dsa_host.o rebalance.o sequencer.o appendmodel.o build.o parbatch.o tables.o pool.o comm.o"
for spec in "$@"; do
  tag="${spec%%:*}"; defs="${spec#*:}"
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c spmv.hip -o /tmp/spmv_$tag.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libdsa_hip_$tag.so $OBJS /tmp/spmv_$tag.o -ldl ) &
done
wait
ls -la libdsa_hip_*.so
