#!/bin/bash
# round 4, visit n: the final resize table (batched / per frame, bilinear + Lanczos-3, RGB / NV12 / YUV420 / Y) and the per-frame Lanczos launch under forced band / strip shapes
mkdir -p gpurun_out
VPF_BENCH_Y=1 timeout 900 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r04n_resize_batch.txt; grep -c resize_batch gpurun_out/r04n_resize_batch.txt
for m in 0 0x801 0x401 0x802 0x402 0x403; do
  echo "forced shape (N-tiles << 8 | tiles per band) $m" >> gpurun_out/r04n_per_frame_shapes.txt
  VPF_BENCH_MFMA=$m VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | grep -E "RGB    (1920x1080->1280x720|3840x2160->1920|1280x720->1920)|NV12   1920" >> gpurun_out/r04n_per_frame_shapes.txt
done
cat gpurun_out/r04n_per_frame_shapes.txt | cut -c1-200 | head -12
