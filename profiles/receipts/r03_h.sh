#!/bin/bash
# round 3, visit h: the driver-style headline five times back to back, the rocprofv3 kernel average of the same command, PMC traffic passes
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
{ echo "# five back-to-back driver-style runs: python bench.py --steps 20 --warmup 5 (--no-cpu for runs 2-5), one MI355X"; 
  for i in 1 2 3 4 5; do
    if [ $i = 1 ]; then python bench.py --steps 20 --warmup 5 2>/dev/null; else python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null; fi | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i: value', d['value'], 'Gpix/s  ms_per_step', d['ms_per_step'], ' per_repeat_ms', d['per_repeat_ms'], ' preheat_ms', d['preheat_ms'], ' roofline.frac', d['roofline']['frac'], ' avg_launch_us', d['roofline']['avg_launch_us'])
if 'cpu_baseline' in d: print('   cpu_baseline', json.dumps(d['cpu_baseline']))"
  done; } > $OUT/r03_driver_style.txt 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/r03_bench_default.json 2>/dev/null
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r03_prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu > "$OUT/r03_prof_bench.json" 2> "$OUT/r03_prof.err"
cd "$GRAFT_REPO_ROOT"
python - <<'PY' >> gpurun_out/r03_driver_style.txt
import csv, glob, json
for f in glob.glob("gpurun_out/r03_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nv12_rgb" in r["Name"]:
            print("rocprofv3 --kernel-trace --stats of the same command:", r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
d = json.loads(open("gpurun_out/r03_prof_bench.json").read().strip().splitlines()[-1])
print("  the profiled process's own line: ms_per_step", d["ms_per_step"], "per_repeat_ms", d["per_repeat_ms"], "avg_launch_us", d["roofline"]["avg_launch_us"])
PY
cat gpurun_out/r03_driver_style.txt
bash tools/gpu_pmc.sh > gpurun_out/r03_gpu_pmc.log 2>&1; tail -12 gpurun_out/pmc/calib_bw.txt
python tools/pmc_summary.py r03 2>&1 | tail -3
