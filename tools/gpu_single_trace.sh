cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/single -o s -- python $GRAFT_REPO_ROOT/bench.py --mode single --steps 20 --warmup 3 --no-cpu --ring 16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/single/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_nv12_rgb" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
g=[int(rows[i+1]["Start_Timestamp"])-int(rows[i]["End_Timestamp"]) for i in range(len(rows)-1)]
g=[x for x in g if x<20000]
import statistics as st
print(rows[0]["Kernel_Name"][:80])
print("n",len(d),"dur median",st.median(d),"mean",sum(d)/len(d),"gap median",st.median(g),"mean",sum(g)/len(g))
PY
