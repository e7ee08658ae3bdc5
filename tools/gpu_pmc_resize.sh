#!/bin/bash
# SQ counters + kernel-trace durations for the resize kernels (tools/pmc_resize_run.py [sw sh dw dh [interp]])
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmc_resize; export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_resize"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_run.py" $@ > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq1 -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_run.py" $@ > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_run.py" $@ > $OUT/sq2.log 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, collections, glob
for d in ("sq1", "sq2"):
    for f in glob.glob(f"gpurun_out/pmc_resize/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "resize" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{d} {k:28s} mean {sum(v)/len(v):14.1f}  n={len(v)}")
for f in glob.glob("gpurun_out/pmc_resize/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resize" in r["Name"]:
            print("trace", r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
