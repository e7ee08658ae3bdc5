#!/bin/bash
# round 3, visit w: output stores of the Lanczos matrix-core kernel: to the picture (0), to one 4-KiB spot per workgroup (64: no HBM write traffic, same instructions), none (128)
mkdir -p gpurun_out
for X in 0 64 128; do timeout 120 python tools/lab/ablate/time_one.py tools/lab/ablate/libvpfhip_x$X.so 2>&1 | grep ablate; done | tee gpurun_out/r03w_ablate.txt
