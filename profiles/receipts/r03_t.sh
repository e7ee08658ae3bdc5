#!/bin/bash
# round 3, visit t: XCD-aware task numbering in the Lanczos matrix-core kernel: tests, timings, HBM traffic
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or policy or fuzz_resize" > gpurun_out/r03t_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03t_pytest.txt
VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03t_bench.txt; cat gpurun_out/r03t_bench.txt
for s in "3840 2160 1920 1080" "1920 1080 1280 720"; do
  n=$(echo $s | tr ' ' '_')
  bash tools/gpu_pmc_resize_traffic.sh $s 2 > gpurun_out/r03t_traffic_$n.txt 2>&1; tail -1 gpurun_out/r03t_traffic_$n.txt
done
