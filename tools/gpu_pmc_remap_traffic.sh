#!/bin/bash
# HBM-side traffic of the remap kernel: FETCH_SIZE / WRITE_SIZE (+ L2 hit/miss) in separate passes, kernel-trace only
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc_remap; export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_remap"
V=${1:-0}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/t_$N -o p -- python "$GRAFT_REPO_ROOT/tools/secondary_bench.py" $V remap > $OUT/t_$N.log 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc_remap/t_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "remap" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:28s} mean {sum(v)/len(v):14.1f}  n={len(v)}")
PY
