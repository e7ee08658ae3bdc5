#!/bin/bash
# round 5, visit e: launch-shape sweep of the matrix-core Lanczos kernel at 128 frames per dispatch (what the planner's cost model must be refitted to),
# same-box A/B of 32 against 128 frames per dispatch on the 4K cases; wave timelines of the Lanczos kernel (debug dump on failure)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python tools/lanczos_shape_sweep.py 128 2 2>&1 | grep lzm-sweep) > $O/r05_e_lanczos_shape_sweep_n128.txt
(SWEEP_Y=1 timeout 400 python tools/lanczos_shape_sweep.py 128 2 2>&1 | grep lzm-sweep) > $O/r05_e_lanczos_shape_sweep_n128_Y.txt
cat $O/r05_e_lanczos_shape_sweep_n128.txt $O/r05_e_lanczos_shape_sweep_n128_Y.txt | cut -c1-600
for nb in 32 128; do echo "== VPF_BENCH_N=$nb"; (VPF_BENCH_N=$nb timeout 300 python tools/resize_batch_bench.py 2>&1 | grep "resize_batch" | grep "3840x2160\|RGB    1920x1080->1280x720" | cut -c1-110); done > $O/r05_e_n32_vs_n128.txt; cat $O/r05_e_n32_vs_n128.txt
for spec in "lanczos Y 1920 1080 1280 720" "lanczos RGB 1920 1080 1280 720" "lanczos RGB 1920 1080 3840 2160"; do
  timeout 120 python tools/wave_times.py $spec 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-400
done > $O/r05_e_wave_times.txt; cat $O/r05_e_wave_times.txt
