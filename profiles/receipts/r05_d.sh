#!/bin/bash
# round 5, visit d: 128 frames per dispatch (kMaxBatch 32 -> 128): the suite, the resize / fused tables, the headline; wave timelines at the product's occupancy
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests -m gpu -q -x --maxfail=5 2>&1 | tail -8) > $O/r05_d_pytest.txt; tail -3 $O/r05_d_pytest.txt
(VPF_BENCH_Y=1 timeout 600 python tools/resize_batch_bench.py 2>&1 | grep resize_batch) > $O/r05_d_resize_batch.txt; cut -c1-150 $O/r05_d_resize_batch.txt
(FUSED_N=128 FUSED_VARIANTS=0,47 timeout 400 python tools/fused_scales_bench.py 2>&1 | grep fused) > $O/r05_d_fused_scales_n128.txt; cut -c1-330 $O/r05_d_fused_scales_n128.txt
timeout 300 python bench.py --no-cpu > $O/r05_d_bench.json 2>/dev/null; cut -c1-400 $O/r05_d_bench.json
for spec in "bilinear Y 1920 1080 1280 720" "bilinear Y 1920 1080 1280 720 --n 128" "lanczos Y 1920 1080 1280 720" "lanczos Y 1920 1080 1280 720 --n 128" "lanczos RGB 1920 1080 1280 720" "lanczos YUV420 1920 1080 1280 720" \
            "lanczos RGB 3840 2160 1920 1080" "lanczos RGB 1920 1080 3840 2160" "lanczos RGB 1280 720 1920 1080"; do
  timeout 120 python tools/wave_times.py $spec 2>&1 | grep -v amdgpu.ids | tail -9
done > $O/r05_d_wave_times.txt; cat $O/r05_d_wave_times.txt
