#!/bin/bash
# round 2, visit M: counters and tables of the band / march kernels for profiles/
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
{ echo "# batched Lanczos, 1920x1080 -> 1280x720 RGB, 32 frames per dispatch (tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 2): the march kernel"; bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 2 2>&1 | tail -26;
  echo; echo "# the same launch on the tiled kernel (VPF_TUNE_RESIZE_MARCH = 1)"; VPF_PMC_MARCH=1 bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 2 2>&1 | tail -26;
  echo; echo "# batched bilinear, same sizes (... 1): the row-band kernel, 4 rows per wave"; bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 1 2>&1 | tail -26;
  echo; echo "# batched bilinear 1920x1080 -> 3840x2160 (... 1): the row-band kernel, 16 rows per wave"; bash tools/gpu_pmc_resize_batch.sh 1920 1080 3840 2160 1 2>&1 | tail -26; } > gpurun_out/r02_pmc_resize_batch.txt
timeout 300 python tools/fused_scales_bench.py > gpurun_out/r02_fused_scales.txt 2>&1
bash tools/gpu_pmc_fused.sh 1920 1080 1280 720 2>&1 | tail -18 > gpurun_out/r02_pmc_fused_1080_720.txt
timeout 300 python tools/lanczos_bench.py > gpurun_out/r02_resize_modes.txt 2>&1
timeout 300 python tools/float_resize_bench.py >> gpurun_out/r02_resize_modes.txt 2>&1
timeout 300 python tools/resize_sizes_bench.py >> gpurun_out/r02_resize_modes.txt 2>&1
grep -E "^trace|^#" gpurun_out/r02_pmc_resize_batch.txt; cat gpurun_out/r02_fused_scales.txt
