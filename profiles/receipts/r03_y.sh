#!/bin/bash
# round 3, visit y: weight tables with the row-table copy by LDS-DMA (no registers): tests, timings at 32 / 8 / 1 frames per dispatch
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or policy or fuzz_resize or graph" > gpurun_out/r03y_pytest.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03y_pytest.txt
VPF_BENCH_ONLY=lanczos VPF_BENCH_ONE=1 timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03y_bench.txt; cat gpurun_out/r03y_bench.txt
VPF_BENCH_MFMA=0x10000 VPF_BENCH_ONLY=lanczos VPF_BENCH_ONE=1 timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03y_bench_notab.txt; cat gpurun_out/r03y_bench_notab.txt
