#!/bin/bash
# round 3, visit aq: MFMA / unpack software pipeline pinned with sched_barrier(0) (sched_group_barrier had let the compiler undo it): tests + same-box A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or fuzz_resize" 2>&1 | tail -2
cp videoprocessingframework_amd/libvpfhip.so /tmp/new.so
for round in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then cp tools/lab/ablate/libvpfhip_prev.so videoprocessingframework_amd/libvpfhip.so; else cp /tmp/new.so videoprocessingframework_amd/libvpfhip.so; fi
    VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch | grep -v 416 | sed "s/^/[$v] /" | cut -c1-150
  done
done | tee gpurun_out/r03aq_ab.txt
cp /tmp/new.so videoprocessingframework_amd/libvpfhip.so
