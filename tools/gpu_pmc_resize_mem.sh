#!/bin/bash
# memory-side counters of the batched resize kernels (tools/pmc_resize_batch_run.py [sw sh dw dh [interp]]): L2 (TCC) requests / hits / misses / HBM
# reads, L1 (TCP) requests and stalls, TA busy — separate passes, kernel-trace only.  Prints whatever counters this rocprofv3 knows of the lists below.
cd "$GRAFT_REPO_ROOT"; OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_mem_$1_$3_$5"; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_TOO_MANY_EA_WRREQS_STALL_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/m$i -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_resize_batch_run.py" $@ > $OUT/m$i.log 2>&1 || tail -3 $OUT/m$i.log
done
cd "$GRAFT_REPO_ROOT"
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/m*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("resize", "plane", "lanczos")):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"mem {k:36s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
