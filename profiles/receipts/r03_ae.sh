#!/bin/bash
# round 3, visit ae: arena-full test; fuzz soak of the resize families on the final Lanczos kernel (VPF_FUZZ_SEEDS=3000: 3000 x 2 tests x 3 cases), whole suite
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "arena or tables" 2>&1 | tail -3
VPF_FUZZ_SEEDS=3000 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "fuzz_resize" > gpurun_out/r03ae_fuzz_soak.txt 2>&1; tail -3 gpurun_out/r03ae_fuzz_soak.txt
timeout 1500 python -m pytest tests -q -x -m gpu > gpurun_out/r03ae_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r03ae_pytest_gpu.txt
