# usage (on the GPU box, from the repo root): bash tools/scripts/pmc_gatherbench2.sh <tag>
# Counter evidence for the x-blocked SpMV variants of tools/gatherbench2.hip (prof mode: a fixed dispatch list).
# Separate --pmc passes as MI355X_MICROARCH.md prescribes; summary -> gpurun_out/<tag>_gatherbench2_pmc.json
set -euo pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
$R/tools/gatherbench2 prof > $R/gpurun_out/${TAG}_gb2_labels.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/${TAG}_gb2_pmc_tcc -o c --output-format csv -- $R/tools/gatherbench2 prof > /dev/null
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_gb2_pmc_fetch -o c --output-format csv -- $R/tools/gatherbench2 prof > /dev/null
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $R/gpurun_out/${TAG}_gb2_pmc_ea -o c --output-format csv -- $R/tools/gatherbench2 prof > /dev/null
rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_gb2_kt -o c --output-format csv -- $R/tools/gatherbench2 prof > /dev/null
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
labels = [l.split(": ", 1)[1].strip() for l in open(f"gpurun_out/{tag}_gb2_labels.txt") if l.startswith("dispatch")]
per = collections.defaultdict(dict)
for d in ("tcc", "fetch", "ea"):
    rows = []
    for f in glob.glob(f"gpurun_out/{tag}_gb2_pmc_{d}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    rank = {i: n for n, i in enumerate(ids)}
    for r in rows:
        n = rank[int(r["Dispatch_Id"])]
        per[n][r["Counter_Name"]] = per[n].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
dur = {}
rows = []
for f in glob.glob(f"gpurun_out/{tag}_gb2_kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for n, r in enumerate(rows):
    dur[n] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = collections.OrderedDict()
for n, lab in enumerate(labels):
    e = out.setdefault(lab, collections.defaultdict(list))
    for k, v in per.get(n, {}).items():
        e[k].append(v)
    if n in dur:
        e["duration_us"].append(dur[n])
summary = {lab: {k: round(sum(v) / len(v), 1) for k, v in e.items()} for lab, e in out.items()}
for lab, e in summary.items():
    h, m = e.get("TCC_HIT_sum", 0), e.get("TCC_MISS_sum", 0)
    if h + m:
        e["l2_hit_rate"] = round(h / (h + m), 4)
    if "FETCH_SIZE" in e:
        e["FETCH_SIZE_MB_raw"] = round(e["FETCH_SIZE"] * 1024 / 1e6, 1) if e["FETCH_SIZE"] < 1e7 else round(e["FETCH_SIZE"] / 1e6, 1)
json.dump({"note": "tools/gatherbench2 prof: 2^24 slots x 12 B (201.3 MB stream), 11.0 M gathers; averages over 3 dispatches; "
                   "FETCH_SIZE raw (KB units as reported by rocprofv3; gfx950 counts 64 B per 128-B streaming request)",
           "variants": summary}, open(f"gpurun_out/{tag}_gatherbench2_pmc.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
