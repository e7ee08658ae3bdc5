#!/bin/bash
# round 3, visit u: Lanczos matrix-core kernel with LDS-DMA staging (three buffers, hand-counted vmcnt): tests, timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or policy or fuzz_resize" > gpurun_out/r03u_pytest.txt 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r03u_pytest.txt
VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03u_bench.txt; cat gpurun_out/r03u_bench.txt
