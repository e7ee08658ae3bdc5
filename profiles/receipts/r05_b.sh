#!/bin/bash
# round 5, visit b: wave timelines with the atomic-free hook (slot = flat wave index) + phase marks in the row-band kernel; the cost of a
# same-address device-scope atomic (work-counter probe)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
./tools/lab/probes/probe_atomic_rate > $O/r05_b_probe_atomic_rate.txt 2>&1; cat $O/r05_b_probe_atomic_rate.txt
for spec in "bilinear Y 1920 1080 1280 720 --band 0" "bilinear Y 1920 1080 1280 720 --band 0x304" "bilinear Y 1920 1080 1280 720 --band 0x504" "bilinear Y 1920 1080 1280 720 --band 0x8" \
            "bilinear NV12 1920 1080 1280 720 --band 0" "bilinear YUV420 1920 1080 1280 720 --band 0" "bilinear RGB 1920 1080 1280 720 --band 0" "lanczos Y 1920 1080 1280 720" "lanczos RGB 1920 1080 1280 720" "lanczos YUV420 1920 1080 1280 720" "lanczos NV12 1920 1080 1280 720" \
            "lanczos RGB 3840 2160 1920 1080" "lanczos RGB 1920 1080 3840 2160" "lanczos RGB 1280 720 1920 1080" "fused NV12 1920 1080 1280 720" "fused NV12 1920 1080 3840 2160"; do
  timeout 120 python tools/wave_times.py $spec 2>&1 | grep "wave_times"
done > $O/r05_b_wave_times.txt; cat $O/r05_b_wave_times.txt
