#!/bin/bash
# round 4, visit y: timing ablations of the row-band bilinear kernel (no blend / no staging / neither) on the 1-, 2- and 3-channel cases
mkdir -p gpurun_out
AB_PASSES=2 timeout 1200 python tools/lab/ablate/time_bl.py tools/lab/ablate/libvpfhip_bl0.so tools/lab/ablate/libvpfhip_bl1.so tools/lab/ablate/libvpfhip_bl2.so tools/lab/ablate/libvpfhip_bl3.so 2>&1 | grep "\[bl\]" | tee gpurun_out/r04y_bilinear_ablate.txt
