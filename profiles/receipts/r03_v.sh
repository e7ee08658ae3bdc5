#!/bin/bash
# round 3, visit v: Lanczos matrix-core kernel, staging loads reshaped to 4 rows x 256 B per instruction: tests, timings, TA counters
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or policy or fuzz_resize" > gpurun_out/r03v_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03v_pytest.txt
VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03v_bench.txt; cat gpurun_out/r03v_bench.txt
bash tools/gpu_pmc_resize_mem.sh 3840 2160 1920 1080 2 > gpurun_out/r03v_mem_4k.txt 2>&1; grep -E "TA_|PENDING|TCC_EA0_RDREQ_sum|TCC_HIT|TCC_MISS" gpurun_out/r03v_mem_4k.txt
