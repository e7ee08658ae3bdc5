#!/bin/bash
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/planar_single -o p -- python $GRAFT_REPO_ROOT/tools/planar_single_trace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, statistics as st
f = glob.glob("gpurun_out/planar_single/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "vpf::" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in acc.items():
    print(f"{k[0]:62s} grid {k[1]:>8s} n {len(v):3d} median {st.median(v)/1e3:6.2f} us  min {min(v)/1e3:6.2f}")
PY
