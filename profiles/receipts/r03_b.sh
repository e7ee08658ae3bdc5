#!/bin/bash
# round 3, visit b: SQ counters of the matrix-core Lanczos kernel, 1080p -> 720p RGB x 32 (policy shape) and 4K -> 1080p
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | head -20
bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 2 > gpurun_out/r03b_pmc_1080_720.txt 2>&1; cat gpurun_out/r03b_pmc_1080_720.txt
tail -3 gpurun_out/pmc_rb_1920_1280_2/sq4.log
bash tools/gpu_pmc_resize_batch.sh 3840 2160 1920 1080 2 > gpurun_out/r03b_pmc_4k_1080.txt 2>&1; cat gpurun_out/r03b_pmc_4k_1080.txt
