#!/bin/bash
# round 4, visit i: does wave priority even out the two co-resident waves of a SIMD (lifetimes spread +-13 % around the mean, the kernel lasts as long as the slowest)?
# x256: the wave in the odd slot runs at priority 1; x512 / x1024: priority alternates between the two every 2 / 8 source tiles
mkdir -p gpurun_out
for rep in 1 2; do
  for x in 0 256 512 1024; do
    timeout 300 python tools/lab/ablate/time_one.py tools/lab/ablate/libvpfhip_x$x.so 2>&1 | grep ablate | tee -a gpurun_out/r04i_ablate.txt
  done
done
