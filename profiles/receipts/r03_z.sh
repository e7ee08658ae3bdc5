#!/bin/bash
# round 3, visit z: XCD-aware block numbering in every batched resize kernel (k_plane_batch / k_planes_mp): tests, bilinear timings, HBM traffic of the up-scale
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "resize or policy or band or tiled" > gpurun_out/r03z_pytest.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03z_pytest.txt
VPF_BENCH_ONLY=bilinear timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03z_bench.txt; cat gpurun_out/r03z_bench.txt
for s in "1920 1080 3840 2160" "1280 720 1920 1080" "1920 1080 1280 720"; do
  n=$(echo $s | tr ' ' '_')
  bash tools/gpu_pmc_resize_traffic.sh $s 1 > gpurun_out/r03z_traffic_$n.txt 2>&1; tail -1 gpurun_out/r03z_traffic_$n.txt
done
