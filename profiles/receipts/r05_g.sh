#!/bin/bash
# round 5, visit g: after the sign-extension fix in the row-table DMA address (visit f crashed on it): the whole suite, the Lanczos table with the refitted planner, wave timelines at 128 frames
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests -m gpu -q -x --maxfail=3 2>&1 | tail -6) > $O/r05_g_pytest.txt; tail -3 $O/r05_g_pytest.txt
(VPF_BENCH_Y=1 timeout 600 python tools/resize_batch_bench.py 2>&1 | grep resize_batch) > $O/r05_g_resize_batch.txt; cut -c1-150 $O/r05_g_resize_batch.txt
for spec in "lanczos Y 1920 1080 1280 720 --n 128" "lanczos NV12 1920 1080 1280 720 --n 128" "lanczos RGB 1920 1080 3840 2160" "bilinear Y 1920 1080 1280 720 --n 128" "bilinear RGB 1920 1080 3840 2160" "fused NV12 1920 1080 3840 2160"; do
  timeout 120 python tools/wave_times.py $spec 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-400
done > $O/r05_g_wave_times.txt; cat $O/r05_g_wave_times.txt
