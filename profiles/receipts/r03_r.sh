#!/bin/bash
# round 3, visit r: where does the Lanczos matrix-core kernel's time go?  Ablated builds (tools/lab/ablate/build.sh: VPF_LZM_X bit 1 no group barriers,
# 2 no output transpose / stores, 4 no pass 2, 8 no staging, 16 no pass 1 arithmetic, 32 setup only), us per frame, 32 frames per dispatch
mkdir -p gpurun_out
for X in 0 1 2 4 8 16 24 32; do timeout 120 python tools/lab/ablate/time_one.py tools/lab/ablate/libvpfhip_x$X.so 2>&1 | grep ablate; done | tee gpurun_out/r03r_ablate.txt
