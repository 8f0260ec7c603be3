# usage (on the GPU box, from the repo root): bash tools/scripts/gather_floor3.sh <tag>
# Runs tools/gatherbench3 (synthetic stream / gather micro-benchmark, sizes on the command line) on the two SpMV shapes bench.py
# reports — config 3 (2^24 slots, 10 M x gathers, 1 M columns) and one config-4 shard (2^25 slots, 12.5 M gathers, 1.25 M columns) —
# and writes gpurun_out/<tag>_gather_floor.json / <tag>_gather_floor_c4shard.json: the slot stream alone, the gathers alone, both in
# one kernel, and the minimal key-driven kernel (one wave per 512 slots, a gather per stored cell, one sum per wave: no semaphores,
# no rows, no y).  Copy them to profiles/gather_floor.json / profiles/gather_floor_c4shard.json to make bench.py quote them.
set -euo pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
mkdir -p $R/gpurun_out
cd $R/tools
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -o gatherbench3 gatherbench3.hip
./gatherbench3 24 10000000 1000000 > $R/gpurun_out/${TAG}_gatherbench3_c3.txt 2>&1
./gatherbench3 25 12500000 1250000 > $R/gpurun_out/${TAG}_gatherbench3_c4shard.txt 2>&1
# the two L2-resident shapes: the banded extra (10 M gathers whose targets sit in a 1 MB stretch of x at any time) and the product on the
# final config-5 matrix (2^21 slots, 800 k cells, x = 50 k doubles = 400 KB)
./gatherbench3 24 10000000 131072 > $R/gpurun_out/${TAG}_gatherbench3_banded.txt 2>&1
./gatherbench3 21 800000 50000 > $R/gpurun_out/${TAG}_gatherbench3_c5.txt 2>&1
cd $R && python3 - <<PY
import hashlib, json, re
sha = hashlib.sha256(open("tools/gatherbench3.hip", "rb").read()).hexdigest()[:16]
for name, out, what in (("c3", "gather_floor", "2^24 slots of 12 B streamed once; 10.0 M 8-byte gathers from an 8.00 MB table (uniformly random indices): the access "
                                                "pattern of k_spmv_gather on config 3 (semaphores do not gather x), without its rows, semaphores and y"),
                        ("c4shard", "gather_floor_c4shard", "2^25 slots of 12 B streamed once; 12.5 M 8-byte gathers from a 10.0 MB table: the slot stream and x gathers of "
                                                            "one config-4 shard (its 80 MB y and 7 M row keys are NOT in this floor)"),
                        ("banded", "gather_floor_banded", "2^24 slots of 12 B streamed once; 10.0 M 8-byte gathers from a 1.0 MB table (every gather an L2 hit): the access "
                                                          "pattern of the banded extra of bench.py, where the x entries a stretch of rows reads sit in a few hundred KB"),
                        ("c5", "gather_floor_c5", "2^21 slots of 12 B streamed once; 0.8 M 8-byte gathers from a 400 KB table: the product on the final config-5 "
                                                  "matrix (100 k rows x 50 k columns, 800 k cells), without its rows, semaphores and y")):
    txt = open("gpurun_out/${TAG}_gatherbench3_%s.txt" % name).read()
    m = re.search(r"grid 1024: stream only ([\d.]+) us \| gather only ([\d.]+) \| both ([\d.]+)", txt)
    k = re.search(r"key-driven.*?gap lanes read x\[0\] ([\d.]+) us \| gap lanes masked off ([\d.]+) us", txt)
    rec = {"stream_only_us": float(m.group(1)), "gather_only_us": float(m.group(2)), "stream_and_gather_one_kernel_us": float(m.group(3)),
           "keyed_single_pass_us": min(float(k.group(1)), float(k.group(2))), "workload": what,
           "source": "tools/gatherbench3.hip (sha256 %s), grid 1024 for the three phases; key-driven kernel: one wave per 512 slots" % sha}
    json.dump(rec, open("gpurun_out/${TAG}_%s.json" % out, "w"), indent=1)
    print(json.dumps(rec))
PY
