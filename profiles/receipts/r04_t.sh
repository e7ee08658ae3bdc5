#!/bin/bash
# round 4, visit t: one dispatch per frame, the matrix-core Lanczos kernel against the tile kernel (VPF_TUNE_RESIZE_MFMA = 1)
mkdir -p gpurun_out
for K in 0 1; do
  echo "== VPF_BENCH_MFMA=$K"
  VPF_BENCH_Y=1 VPF_BENCH_MFMA=$K VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch" | cut -c1-220
done | tee gpurun_out/r04t_single_frame_kernels.txt
