#!/bin/bash
# round 6, visit k: 64 frames per dispatch for mid-sized bilinear frames (7-10 MB: packed RGB 1080p <-> 720p) against 32, same box, sustained protocol
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
for rep in 1 2; do
  for M in 0 1; do
    (VPF_HIP_MID_BATCH=$M VPF_BENCH_ONLY=bilinear timeout 600 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch.*RGB" | grep "1280x720\|1920x1080->1280" | sed "s/^/[mid=$M] /") >> $O/r06_k_mid_batch_ab.txt
  done
done
cut -c1-200 $O/r06_k_mid_batch_ab.txt
for rep in 1 2; do
  for M in 0 1; do
    (VPF_HIP_MID_BATCH=$M FUSED_N=64 FUSED_PAIR=0 timeout 600 python tools/fused_scales_bench.py 2>&1 | grep "fused" | grep "1280x720 -> 1920x1080" | sed "s/^/[mid=$M, 64 frames per call] /") >> $O/r06_k_mid_batch_fused_ab.txt
  done
done
cut -c1-220 $O/r06_k_mid_batch_fused_ab.txt
