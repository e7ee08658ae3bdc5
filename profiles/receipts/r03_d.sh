#!/bin/bash
# round 3, visit d: SQ counters of the matrix-core Lanczos kernel, 1080p -> 720p RGB x 32
mkdir -p gpurun_out
bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 2 > gpurun_out/r03d_pmc_1080_720.txt 2>&1; cat gpurun_out/r03d_pmc_1080_720.txt
