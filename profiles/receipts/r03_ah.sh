#!/bin/bash
# round 3, visit ah: same-box A/B of the gather Lanczos kernel (byte loads vs window loads) on the sample chain and per-frame dispatch
mkdir -p gpurun_out
cp videoprocessingframework_amd/libvpfhip.so /tmp/new.so
for round in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then cp tools/lab/ablate/libvpfhip_prev.so videoprocessingframework_amd/libvpfhip.so; else cp /tmp/new.so videoprocessingframework_amd/libvpfhip.so; fi
    timeout 300 python tools/chain_bench.py 2>&1 | grep "lanczos3" | sed "s/^/[$v] /" | cut -c1-330
    VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep -E "416x416" | sed "s/^/[$v] /" | cut -c1-220
  done
done | tee gpurun_out/r03ah_ab.txt
cp /tmp/new.so videoprocessingframework_amd/libvpfhip.so
