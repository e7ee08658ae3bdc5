#!/bin/bash
# round 4, visit l: the round's Lanczos kernel under the counters (SQ passes, HBM traffic) at 1080p->720p and 4K->1080p, and the whole batched / per-frame resize table
mkdir -p gpurun_out
for s in "1920 1080 1280 720" "3840 2160 1920 1080"; do
  n=$(echo $s | tr ' ' '_')
  bash tools/gpu_pmc_resize_batch.sh $s 2 > gpurun_out/r04l_pmc_lanczos_$n.txt 2>&1; tail -3 gpurun_out/r04l_pmc_lanczos_$n.txt | cut -c1-200
  bash tools/gpu_pmc_resize_traffic.sh $s 2 > gpurun_out/r04l_traffic_lanczos_$n.txt 2>&1; tail -1 gpurun_out/r04l_traffic_lanczos_$n.txt | cut -c1-300
done
VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r04l_resize_batch.txt; cat gpurun_out/r04l_resize_batch.txt | cut -c1-260
