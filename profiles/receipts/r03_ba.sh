#!/bin/bash
# round 3, visit ba: band walk with role-swapping lerp registers and lane-computed row taps — parity of the band / fused families, bilinear + fused timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "band or resize or fused or fuzz" 2>&1 | tail -3 | tee gpurun_out/r03ba_pytest.txt
VPF_BENCH_ONLY=bilinear VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | tee gpurun_out/r03ba_resize_batch_bilinear.txt
timeout 600 python tools/fused_scales_bench.py 2>&1 | grep fused | tee gpurun_out/r03ba_fused_scales.txt
