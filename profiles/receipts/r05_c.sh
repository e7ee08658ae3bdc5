#!/bin/bash
# round 5, visit c: wave timelines without the phase marks (the marks cost the row-band kernel a wave per SIMD), Lanczos timelines, kernarg size probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
./tools/lab/probes/probe_kernarg_size > $O/r05_c_probe_kernarg_size.txt 2>&1; cat $O/r05_c_probe_kernarg_size.txt
for spec in "bilinear Y 1920 1080 1280 720 --band 0" "bilinear Y 1920 1080 1280 720 --band 0x304" "bilinear Y 1920 1080 1280 720 --band 0x504" "bilinear Y 1920 1080 1280 720 --band 0x104" \
            "bilinear NV12 1920 1080 1280 720 --band 0" "bilinear YUV420 1920 1080 1280 720 --band 0" "lanczos Y 1920 1080 1280 720" "lanczos RGB 1920 1080 1280 720" "lanczos YUV420 1920 1080 1280 720" "lanczos NV12 1920 1080 1280 720" \
            "lanczos RGB 3840 2160 1920 1080" "lanczos RGB 1920 1080 3840 2160" "lanczos RGB 1280 720 1920 1080"; do
  timeout 120 python tools/wave_times.py $spec 2>&1 | tail -12
done > $O/r05_c_wave_times.txt; cat $O/r05_c_wave_times.txt
