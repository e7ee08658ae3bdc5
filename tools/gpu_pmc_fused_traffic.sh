#!/bin/bash
# HBM-side traffic of the fused NV12 -> resize -> RGB kernel (tools/pmc_fused_run.py [sw sh dw dh], 32 frames per dispatch): FETCH_SIZE x2 / WRITE_SIZE, separate passes
cd "$GRAFT_REPO_ROOT"; OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_ft_$1_$3"; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/t_$C -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_fused_run.py" $@ > $OUT/t_$C.log 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - "$OUT" $@ <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
sw, sh, dw, dh = (int(a) for a in sys.argv[2:6])
acc = collections.defaultdict(list); name = None; n = None
for f in sorted(glob.glob(f"{out}/t_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "convert" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); name = r["Kernel_Name"]; n = int(r.get("Grid_Size_Z", 0) or 0) or n
rd = 2.0 * 1024 * sum(acc["FETCH_SIZE"]) / max(1, len(acc["FETCH_SIZE"]))
wr = 1024.0 * sum(acc["WRITE_SIZE"]) / max(1, len(acc["WRITE_SIZE"]))
frames = 32
alg_r, alg_w = frames * sw * sh * 3 // 2, frames * 3 * dw * dh
print(f"{(name or '?')[:70]}  NV12 {sw}x{sh} -> RGB {dw}x{dh} x {frames} frames: HBM read {rd / 1e6:8.1f} MB (whole source {alg_r / 1e6:8.1f}: x{rd / alg_r:.3f})  "
      f"written {wr / 1e6:8.1f} MB (destination {alg_w / 1e6:8.1f}: x{wr / alg_w:.3f})  total / (source + destination) = {(rd + wr) / (alg_r + alg_w):.3f}")
PY
