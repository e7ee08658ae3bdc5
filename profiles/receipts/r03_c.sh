#!/bin/bash
# round 3, visit c: Lanczos parity tests + policy-shape timings after the LDS opt-in (4K -> 1080p on 8-tile strips)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or policy or fuzz_resize" > gpurun_out/r03c_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03c_pytest.txt
VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03c_bench.txt; cat gpurun_out/r03c_bench.txt
