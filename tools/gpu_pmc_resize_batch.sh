#!/bin/bash
# SQ counters + kernel-trace durations of the batched resize kernels (tools/pmc_resize_batch_run.py [sw sh dw dh [interp]])
cd "$GRAFT_REPO_ROOT"; OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_rb_$1_$3_$5"; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_batch_run.py" $@ > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq1 -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_batch_run.py" $@ > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_batch_run.py" $@ > $OUT/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/sq3 -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_batch_run.py" $@ > $OUT/sq3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/sq4 -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_batch_run.py" $@ > $OUT/sq4.log 2>&1
cd "$GRAFT_REPO_ROOT"
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
for d in ("sq1", "sq2", "sq3", "sq4"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "resize" in r["Kernel_Name"] or "plane" in r["Kernel_Name"] or "lanczos" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{d} {k:28s} mean {sum(v)/len(v):14.1f}  n={len(v)}")
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resize" in r["Name"] or "plane" in r["Name"] or "lanczos" in r["Name"]:
            print("trace", r["Name"][:90], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
