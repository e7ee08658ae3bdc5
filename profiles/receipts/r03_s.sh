#!/bin/bash
# round 3, visit s: HBM traffic (FETCH_SIZE / WRITE_SIZE) and L1 / L2 counters of the Lanczos matrix-core kernel
mkdir -p gpurun_out
for s in "3840 2160 1920 1080" "1920 1080 1280 720"; do
  n=$(echo $s | tr ' ' '_')
  bash tools/gpu_pmc_resize_traffic.sh $s 2 > gpurun_out/r03s_traffic_$n.txt 2>&1; tail -2 gpurun_out/r03s_traffic_$n.txt
  bash tools/gpu_pmc_resize_mem.sh $s 2 > gpurun_out/r03s_mem_$n.txt 2>&1; grep "^mem" gpurun_out/r03s_mem_$n.txt
done
grep -E "TCC_|TCP_|TA_" gpurun_out/pmc_mem_3840_1920_2/counters.txt | grep -oE "(TCC|TCP|TA)_[A-Z0-9_a-z]+" | sort -u | tr '\n' ' ' | head -c 6000
