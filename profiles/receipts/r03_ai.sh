#!/bin/bash
# round 3, visit ai: gather Lanczos with the window path chosen per wave: tests + same-box A/B against the byte-load version
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or fuzz_resize or resize_batch" 2>&1 | tail -2
cp videoprocessingframework_amd/libvpfhip.so /tmp/new.so
for round in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then cp tools/lab/ablate/libvpfhip_prev.so videoprocessingframework_amd/libvpfhip.so; else cp /tmp/new.so videoprocessingframework_amd/libvpfhip.so; fi
    timeout 300 python tools/chain_bench.py 2>&1 | grep "lanczos3" | sed "s/^/[$v] /" | cut -c90-330
    VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep -E "416x416" | sed "s/^/[$v] /" | cut -c1-220
  done
done | tee gpurun_out/r03ai_ab.txt
cp /tmp/new.so videoprocessingframework_amd/libvpfhip.so
