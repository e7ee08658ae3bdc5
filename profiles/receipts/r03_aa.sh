#!/bin/bash
# round 3, visit aa: XCD-aware block numbering in the fused convert+resize batch kernels: tests, timings across scale factors, HBM traffic
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or convert_resize" > gpurun_out/r03aa_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03aa_pytest.txt
timeout 300 python tools/fused_scales_bench.py 2>&1 | grep fused > gpurun_out/r03aa_fused.txt; cat gpurun_out/r03aa_fused.txt
for s in "1920 1080 3840 2160" "1920 1080 1280 720"; do
  n=$(echo $s | tr ' ' '_')
  bash tools/gpu_pmc_fused_traffic.sh $s > gpurun_out/r03aa_traffic_$n.txt 2>&1; tail -2 gpurun_out/r03aa_traffic_$n.txt
done
