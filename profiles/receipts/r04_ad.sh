#!/bin/bash
# round 4, visit ad: same-box A/B of the windowed tap reads (one 8-byte LDS window + v_perm_b32 per pixel pair / pixel) on 1- and 2-channel planes
mkdir -p gpurun_out
AB_PASSES=4 timeout 900 python tools/lab/ablate/time_bl.py tools/lab/ablate/libvpfhip_blnowin.so tools/lab/ablate/libvpfhip_bl0.so 2>&1 | grep "\[bl\]" | tee gpurun_out/r04ad_windowed_ab.txt
