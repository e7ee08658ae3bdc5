#!/bin/bash
# round 3, visit ad: one frame per dispatch on the matrix-core Lanczos kernel: forced shapes with one and two tiles per band, and rocprofv3 kernel durations / gaps of the per-frame loop
mkdir -p gpurun_out
for shape in 0 0x401 0x402 0x403 0x801 0x802; do
  VPF_BENCH_MFMA=$shape VPF_BENCH_ONLY=lanczos timeout 200 python tools/resize_batch_bench.py 2>&1 | grep -E "RGB    1920x1080->1280x720|NV12   1920x1080->1280x720|RGB    1280x720->1920x1080" | sed "s/^/[shape $shape] /" | cut -c1-200
done | tee gpurun_out/r03ad_single.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03ad_trace -o t -- python $GRAFT_REPO_ROOT/tools/pmc_resize_run.py 1920 1080 1280 720 2 > $GRAFT_REPO_ROOT/gpurun_out/r03ad_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r03ad_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("stats", r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
for f in glob.glob("gpurun_out/r03ad_trace/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "lanczos" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    if rows:
        gaps.sort(); durs.sort()
        print(f"per-frame launches: n {len(rows)}  duration median {durs[len(durs)//2]} ns  gap median {gaps[len(gaps)//2]} ns  grid {rows[-1].get('Grid_Size_X','?')}x{rows[-1].get('Grid_Size_Y','?')} wg {rows[-1].get('Workgroup_Size_X','?')}")
PY
