#!/bin/bash
# round 3, visit av: fuzz soak of all four families on the final code, VPF_FUZZ_SEEDS=12000 (48 000 tests)
mkdir -p gpurun_out
VPF_FUZZ_SEEDS=12000 timeout 3000 python -m pytest tests/test_gpu_parity.py -q -x -n 6 -k "fuzz" > gpurun_out/r03av_fuzz_soak_big.txt 2>&1; tail -3 gpurun_out/r03av_fuzz_soak_big.txt
