#!/bin/bash
# round 3, visit bh: same-box A/B of the library before (commit 2af74d1) and after the pack / band-walk / exact-LDS-rows changes, alternating, three rounds
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python tools/lab/ab/time_lib.py tools/lab/ab/libvpfhip_before_pack.so 2>&1 | grep "\[ab\]"
  timeout 300 python tools/lab/ab/time_lib.py videoprocessingframework_amd/libvpfhip.so 2>&1 | grep "\[ab\]"
done | tee gpurun_out/r03_ab_before_after_pack.txt
