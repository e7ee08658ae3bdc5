#!/bin/bash
# round 5, visit f: the refitted Lanczos planner at 128 frames per dispatch; bilinear band-knob sweep at 128 frames per dispatch; Lanczos wave timelines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lanczos or policy or persistent" 2>&1 | tail -3)
(VPF_BENCH_Y=1 VPF_BENCH_ONLY=lanczos timeout 600 python tools/resize_batch_bench.py 2>&1 | grep resize_batch) > $O/r05_f_resize_batch_lanczos.txt; cut -c1-150 $O/r05_f_resize_batch_lanczos.txt
(SWEEP_N=128 timeout 400 python tools/band_knob_sweep.py 0 2 4 8 16 0x104 0x204 0x304 0x404 0x604 2>&1 | grep knobs) > $O/r05_f_band_knobs_n128.txt; cat $O/r05_f_band_knobs_n128.txt
for spec in "lanczos Y 1920 1080 1280 720" "lanczos Y 1920 1080 1280 720 --n 128" "lanczos RGB 1920 1080 1280 720" "lanczos RGB 1920 1080 3840 2160" "lanczos RGB 3840 2160 1920 1080"; do
  timeout 120 python tools/wave_times.py $spec 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-400
done > $O/r05_f_wave_times.txt; cat $O/r05_f_wave_times.txt
