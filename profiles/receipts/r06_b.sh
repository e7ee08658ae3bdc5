#!/bin/bash
# round 6, visit b: SQ counters of the two kernels VERDICT r5 item 3 names (bilinear RGB 1080p -> 720p batched, fused NV12 -> RGB 1080p -> 720p) for the blend's ISA budget
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
export PYTHONPATH=$PWD
bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 1 > $O/r06_b_pmc_bilinear_rgb_1080_720.txt 2>&1; cat $O/r06_b_pmc_bilinear_rgb_1080_720.txt
bash tools/gpu_pmc_fused.sh 1920 1080 1280 720 > $O/r06_b_pmc_fused_1080_720.txt 2>&1; cat $O/r06_b_pmc_fused_1080_720.txt
bash tools/gpu_pmc_resize_batch.sh 1920 1080 3840 2160 1 > $O/r06_b_pmc_bilinear_rgb_1080_4k.txt 2>&1; cat $O/r06_b_pmc_bilinear_rgb_1080_4k.txt
