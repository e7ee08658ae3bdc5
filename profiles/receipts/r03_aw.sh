#!/bin/bash
# round 3, visit aw: bilinear on 1- and 2-channel planes: Y alone, NV12, YUV420 at 1080p->720p and 4K->1080p, which kernels run (VPF_HIP_LOG=2), forced rows per wave
mkdir -p gpurun_out
VPF_HIP_LOG=2 VPF_BENCH_Y=1 VPF_BENCH_ONLY=bilinear timeout 300 python tools/resize_batch_bench.py 2>&1 | grep -E "resize_batch|k_planes|k_plane|RowBand|RowPair" | grep -E "NV12|YUV420| Y |RowBand|RowPair|planes" | awk '!seen[$0]++' | head -40 | cut -c1-200
for band in 4 8 16; do VPF_BENCH_BAND=$band VPF_BENCH_Y=1 VPF_BENCH_ONLY=bilinear timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | grep -E "NV12|YUV420| Y " | sed "s/^/[band $band] /" | cut -c1-150; done
