#!/bin/bash
# round 3, visit l: SQ counters of the matrix-core Lanczos kernel at 4K -> 1080p and 1080p -> 720p
mkdir -p gpurun_out
bash tools/gpu_pmc_resize_batch.sh 3840 2160 1920 1080 2 > gpurun_out/r03l_pmc_4k_1080.txt 2>&1; grep -E "sq1|sq2 SQ_(WAIT|ACTIVE_INST_ANY)|sq4 SQ_(INSTS_MFMA|VALU_MFMA_BUSY)|trace" gpurun_out/r03l_pmc_4k_1080.txt
bash tools/gpu_pmc_resize_batch.sh 1920 1080 1280 720 2 > gpurun_out/r03l_pmc_1080_720.txt 2>&1; grep -E "sq1|sq2 SQ_(WAIT|ACTIVE_INST_ANY)|sq4 SQ_(INSTS_MFMA|VALU_MFMA_BUSY)|trace" gpurun_out/r03l_pmc_1080_720.txt
grep -E "LDS_Block_Size|Scratch|VGPR|Kernel_Name" gpurun_out/pmc_rb_3840_1920_2/trace/t_kernel_trace.csv | head -3; head -2 gpurun_out/pmc_rb_3840_1920_2/trace/t_kernel_trace.csv | cut -c1-600
