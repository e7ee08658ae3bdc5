#!/bin/bash
# round 3, visit ac: whole GPU suite on the refit planner + pin-kit rehearsal with quirk cases; Lanczos timings; counters of the final kernel
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r03ac_pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r03ac_pytest_gpu.txt
VPF_BENCH_ONLY=lanczos VPF_BENCH_ONE=1 timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03ac_bench.txt; cat gpurun_out/r03ac_bench.txt
for s in "1920 1080 1280 720" "3840 2160 1920 1080"; do
  n=$(echo $s | tr ' ' '_')
  bash tools/gpu_pmc_resize_batch.sh $s 2 > gpurun_out/r03ac_sq_$n.txt 2>&1; grep -E "sq|trace" gpurun_out/r03ac_sq_$n.txt | grep -E "VALU |INSTS_VALU|MFMA|WAVE|WAIT_ANY|BUSY_CYCLES|LDS_BANK|trace"
  bash tools/gpu_pmc_resize_traffic.sh $s 2 > gpurun_out/r03ac_traffic_$n.txt 2>&1; tail -1 gpurun_out/r03ac_traffic_$n.txt
done
