cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r06_t10
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/scripts/run_case.py tripolar_land 2048 on 3 10 > $OUT/trace.txt 2>&1
python3 $R/scripts/summarize_structure_profile.py $OUT | head -30
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
for r in rows: r["s"],r["e"]=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
rows.sort(key=lambda r:r["s"])
# last ~80 launches timeline
t0=rows[-80]["s"]
for r in rows[-80:]:
    print(f'{(r["s"]-t0)/1e3:9.1f} {(r["e"]-r["s"])/1e3:8.1f}  {r["Kernel_Name"].split("(")[0][:70]}')
PY
