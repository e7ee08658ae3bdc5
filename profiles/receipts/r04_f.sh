#!/bin/bash
# round 4, visit f: does breaking the chip-wide lockstep of the march help?  (x64 / x128: workgroup-dependent start delays of up to 15 x 512 / 2048 cycles)
mkdir -p gpurun_out
for rep in 1 2; do
  for x in 0 64 128; do
    timeout 300 python tools/lab/ablate/time_one.py tools/lab/ablate/libvpfhip_x$x.so 2>&1 | grep ablate | tee -a gpurun_out/r04f_ablate.txt
  done
done
