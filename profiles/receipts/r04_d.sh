#!/bin/bash
# round 4, visit d: timing ablations of the latency-hiding Lanczos kernel (x1: three MFMAs per tile in pass 2; x2: setup only; x4: no pass 2; x8: no pass-1 arithmetic; x12: staging + stores only)
mkdir -p gpurun_out
for rep in 1 2; do
  for x in 0 1 2 4 8 12; do
    timeout 300 python tools/lab/ablate/time_one.py tools/lab/ablate/libvpfhip_x$x.so 2>&1 | grep ablate | tee -a gpurun_out/r04d_ablate.txt
  done
done
