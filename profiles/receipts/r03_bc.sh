#!/bin/bash
# round 3, visit bc: band height forced to 2 / 4 / 8 / 16 after the LDS rows became exact (occupancy vs shared work)
mkdir -p gpurun_out
for b in 0 2 4 8 16; do echo "== VPF_BENCH_BAND=$b"; VPF_BENCH_BAND=$b VPF_BENCH_ONLY=bilinear VPF_BENCH_Y=1 timeout 400 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | cut -c1-112; done | tee gpurun_out/r03bc_band_heights.txt
