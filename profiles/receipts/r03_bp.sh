#!/bin/bash
# round 3, visit bp: rocprofv3 kernel time of the batched resizes with a working set past the Infinity Cache (the runner used to re-dispatch one 287-MB batch)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for args in "1920 1080 1280 720 1" "1920 1080 1280 720 2" "3840 2160 1920 1080 2" "1920 1080 3840 2160 1"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/bp_$(echo $args | tr ' ' '_'); rm -rf $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/tools/pmc_resize_batch_run.py $args > /dev/null 2>&1
  python - "$OUT" "$args" <<'PY'
import csv, glob, sys
out, args = sys.argv[1], sys.argv[2]
for f in glob.glob(f"{out}/**/*kernel_trace.csv", recursive=True):
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    mine = [(i, k) for i, k in enumerate(ks) if "planes_mp" in k[2] or "lanczos_mfma" in k[2] or "plane_batch" in k[2]]
    durs = [(k[1] - k[0]) / 1e3 for _, k in mine][2:]
    gaps = [(ks[i][0] - ks[i - 1][1]) / 1e3 for i, k in mine if i > 0 and ks[i - 1][2] == k[2]][1:]
    print(f"[kernel-time] {args}: {mine[0][1][2][:60]} n={len(durs)} median duration {sorted(durs)[len(durs)//2]:.1f} us per 32 frames = {sorted(durs)[len(durs)//2]/32:.2f} us/frame | median gap to the previous dispatch {sorted(gaps)[len(gaps)//2]:.1f} us")
PY
done | tee $GRAFT_REPO_ROOT/gpurun_out/r03_kernel_time_past_cache.txt
