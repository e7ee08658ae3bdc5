#!/bin/bash
# round 3, visit an: same-box A/B of the XCD-aware block numbering on the short-wave kernels (bilinear row-band, fused strip): does the renumbering arithmetic cost time?
mkdir -p gpurun_out
cp videoprocessingframework_amd/libvpfhip.so /tmp/new.so
for round in 1 2; do
  for v in noxcd new; do
    if [ $v = noxcd ]; then cp tools/lab/ablate/libvpfhip_noxcd.so videoprocessingframework_amd/libvpfhip.so; else cp /tmp/new.so videoprocessingframework_amd/libvpfhip.so; fi
    VPF_BENCH_ONLY=bilinear timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | sed "s/^/[$v] /" | cut -c1-140
    timeout 300 python tools/fused_scales_bench.py 2>&1 | grep fused | sed "s/^/[$v] /"
  done
done | tee gpurun_out/r03an_ab.txt
cp /tmp/new.so videoprocessingframework_amd/libvpfhip.so
