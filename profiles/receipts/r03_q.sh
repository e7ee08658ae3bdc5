#!/bin/bash
# round 3, visit q: Lanczos matrix-core kernel with software-pipelined MFMA / unpack, SDWA pack, 32-bit plane offsets
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lanczos or mfma or policy or fuzz_resize" > gpurun_out/r03q_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03q_pytest.txt
VPF_BENCH_ONLY=lanczos timeout 300 python tools/resize_batch_bench.py 2>&1 | grep resize_batch > gpurun_out/r03q_bench.txt; cat gpurun_out/r03q_bench.txt
