#!/bin/bash
# round 3, visit am: tiled Lanczos test (forced shapes, matrix-core kernel on / off), whole suite, fuzz soak of the resize families, Lanczos table
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos_tile" 2>&1 | tail -3
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r03am_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r03am_pytest_gpu.txt
VPF_FUZZ_SEEDS=3000 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "fuzz_resize" > gpurun_out/r03am_fuzz_soak.txt 2>&1; tail -2 gpurun_out/r03am_fuzz_soak.txt
VPF_BENCH_ONLY=lanczos VPF_BENCH_ONE=1 timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03am_bench.txt; cat gpurun_out/r03am_bench.txt | cut -c1-250
