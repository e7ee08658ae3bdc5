#!/bin/bash
# round 3, visit p: HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and SQ counters of the matrix-core Lanczos kernel, final shapes
mkdir -p gpurun_out
for s in "1920 1080 1280 720" "3840 2160 1920 1080"; do
  bash tools/gpu_pmc_resize_traffic.sh $s 2 > gpurun_out/r03p_traffic_$(echo $s | tr ' ' '_').txt 2>&1; tail -12 gpurun_out/r03p_traffic_$(echo $s | tr ' ' '_').txt
  bash tools/gpu_pmc_resize_batch.sh $s 2 > gpurun_out/r03p_sq_$(echo $s | tr ' ' '_').txt 2>&1; grep -E "sq|trace" gpurun_out/r03p_sq_$(echo $s | tr ' ' '_').txt
done
