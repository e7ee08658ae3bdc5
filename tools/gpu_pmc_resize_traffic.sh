#!/bin/bash
# HBM-side traffic of the batched resize kernels (tools/pmc_resize_batch_run.py [sw sh dw dh [interp]], 32 frames per dispatch): FETCH_SIZE / WRITE_SIZE in
# separate passes, kernel-trace only.  gfx950 corrections as calibrated by tools/pmc_calib.hip (profiles/r02_pmc_calib_bw.txt): FETCH_SIZE counts 128-B
# read requests at 64 B (x2), WRITE_SIZE is exact; both in KiB.
cd "$GRAFT_REPO_ROOT"; OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_rt_$1_$3_$5"; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/t_$C -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_batch_run.py" $@ > $OUT/t_$C.log 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - "$OUT" $@ <<'PY'
import csv, collections, glob, os, sys
out = sys.argv[1]
sw, sh, dw, dh = (int(a) for a in sys.argv[2:6])
ch = 1 if os.environ.get("VPF_PMC_FMT") == "Y" else 3
acc = collections.defaultdict(list)
name = None
for f in sorted(glob.glob(f"{out}/t_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("resize", "plane", "lanczos")):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); name = r["Kernel_Name"]
rd = 2.0 * 1024 * sum(acc["FETCH_SIZE"]) / max(1, len(acc["FETCH_SIZE"]))
wr = 1024.0 * sum(acc["WRITE_SIZE"]) / max(1, len(acc["WRITE_SIZE"]))
alg_r, alg_w = 32 * ch * sw * sh, 32 * ch * dw * dh
print(f"{(name or '?')[:80]}  {sw}x{sh} -> {dw}x{dh} x 32 frames: HBM read {rd / 1e6:8.1f} MB (source bytes {alg_r / 1e6:8.1f}: x{rd / alg_r:.3f})  "
      f"written {wr / 1e6:8.1f} MB (destination bytes {alg_w / 1e6:8.1f}: x{wr / alg_w:.3f})  total / algorithmic = {(rd + wr) / (alg_r + alg_w):.3f}")
PY
