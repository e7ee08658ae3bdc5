#!/bin/bash
# round 4, visit e: timing ablations of the memory side of the Lanczos kernel (x16: no global loads; x32: no global stores; x48: neither = arithmetic + LDS only)
mkdir -p gpurun_out
for rep in 1 2; do
  for x in 0 16 32 48; do
    timeout 300 python tools/lab/ablate/time_one.py tools/lab/ablate/libvpfhip_x$x.so 2>&1 | grep ablate | tee -a gpurun_out/r04e_ablate.txt
  done
done
