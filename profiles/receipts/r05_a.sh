#!/bin/bash
# round 5, visit a: the suite on the round's first changes (workspace record fix, workgroup-shared fused strips, persistent band launch), the
# baseline tables of this box, band-knob sweep (march depth 3..6, persistent forms), wave timelines, fused-vs-pair table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests -m gpu -q -x --maxfail=5 2>&1 | tail -25) > $O/r05_a_pytest.txt
tail -3 $O/r05_a_pytest.txt
(FUSED_VARIANTS=0,47,48 timeout 400 python tools/fused_scales_bench.py 2>&1) > $O/r05_a_fused_scales.txt; tail -9 $O/r05_a_fused_scales.txt
(timeout 300 python tools/band_knob_sweep.py 0 0x204 0x304 0x404 0x504 0x604 0x804 0x10104 0x10204 0x10304 0x10008 2>&1) > $O/r05_a_band_knobs.txt; cat $O/r05_a_band_knobs.txt
for spec in "bilinear Y 1920 1080 1280 720 --band 0" "bilinear Y 1920 1080 1280 720 --band 0x504" "bilinear Y 1920 1080 1280 720 --band 0x10104" "bilinear Y 1920 1080 1280 720 --band 0x10204" \
            "bilinear NV12 1920 1080 1280 720 --band 0" "bilinear RGB 1920 1080 1280 720 --band 0" "lanczos Y 1920 1080 1280 720" "lanczos RGB 1920 1080 1280 720" "lanczos YUV420 1920 1080 1280 720" \
            "lanczos RGB 3840 2160 1920 1080" "lanczos RGB 1920 1080 3840 2160" "fused NV12 1920 1080 1280 720" "fused NV12 1920 1080 1280 720 --variant 47" "fused NV12 1920 1080 3840 2160"; do
  timeout 120 python tools/wave_times.py $spec 2>&1 | grep "wave_times"
done > $O/r05_a_wave_times.txt; cat $O/r05_a_wave_times.txt
(VPF_BENCH_Y=1 timeout 600 python tools/resize_batch_bench.py 2>&1 | grep resize_batch) > $O/r05_a_resize_batch.txt; cut -c1-150 $O/r05_a_resize_batch.txt
