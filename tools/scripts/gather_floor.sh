# usage (on the GPU box, from the repo root): bash tools/scripts/gather_floor.sh <tag>
# Runs tools/gatherbench2 (the synthetic stream / gather micro-benchmark of the SpMV access pattern: 2^24 slots of 12 B, 11 M gathers
# from an 8.4 MB table) and writes the three numbers bench.py quotes beside roofline.frac — the stream alone, the gathers alone, both
# in one kernel — to gpurun_out/<tag>_gather_floor.json (copy it to profiles/gather_floor.json to make bench.py use it).
set -euo pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
mkdir -p $R/gpurun_out
cd $R/tools
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gatherbench2 gatherbench2.hip
./gatherbench2 > $R/gpurun_out/${TAG}_gatherbench2.txt 2>&1
cd $R && python3 - <<PY
import hashlib, json, re
txt = open("gpurun_out/${TAG}_gatherbench2.txt").read()
sec = txt[txt.index("== E5 roles, x table 8.39 MB"):]
m = re.search(r"grid 1024: every wave both queues: stream only ([\d.]+) us \| gather only ([\d.]+) \| both ([\d.]+)", sec)
e1 = re.search(r"== x 8.00 MB.*?E1 fused, one span per wave: nt ([\d.]+) us", txt, re.S)
out = {"stream_only_us": float(m.group(1)), "gather_only_us": float(m.group(2)), "stream_and_gather_one_kernel_us": float(m.group(3)),
       "fused_single_pass_us": float(e1.group(1)) if e1 else None,
       "workload": "2^24 slots of 12 B streamed once; 11.0 M 8-byte gathers from an 8.39 MB table (uniformly random indices): the access pattern "
                   "of k_spmv_gather on config 3, without its arithmetic",
       "source": "tools/gatherbench2.hip (sha256 %s), E5 'every wave both queues', grid 1024" % hashlib.sha256(open("tools/gatherbench2.hip", "rb").read()).hexdigest()[:16]}
json.dump(out, open("gpurun_out/${TAG}_gather_floor.json", "w"), indent=1)
print(json.dumps(out))
PY
