for B in 64 128 256 512; do
DSA_META_BLOCKS=$B bash tools/scripts/trace_tool.sh meta_b$B prof_spmv.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
ts=[]
for f in glob.glob("gpurun_out/meta_b$B"+"_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "meta" in r["Kernel_Name"]: ts.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("blocks", $B, ts)
PY
done
